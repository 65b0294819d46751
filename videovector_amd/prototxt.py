"""Writes net / solver prototxts with the graph structure of the reference's
projects/videovec_embedding/mednet_embedding_train.prototxt (TRAIN phase) for any batch size,
context size, negative count and embedding width -- the shipped file is hand-unrolled for
num_negative_samples = 10; BASELINE configs need 2, 50 and 200.  Layer and blob names follow the
shipped file so that snapshots and tools keyed on names (fc7, ip2, target_score, ...) carry over."""


def _test_branch_front(a, test_source, test_batch, frames):
    """TEST-phase layers in front of fc7 (mednet_embedding_train.prototxt:30-45,75-104,133-177)."""
    a('layers {\n  name: "shot_windows"\n  type: VIDEO_SHOT_WINDOW_TEST_DATA\n  top: "data"\n  top: "video_ids"\n'
      '  video_shot_window_test_data_param {\n    source: "%s"\n    backend: LMDB\n    batch_size: %d\n  }\n'
      '  include: { phase: TEST }\n}' % (test_source, test_batch))
    d = ["context_datum_%d" % j for j in range(1, frames + 1)]
    a('layers {\n  name: "slice_input_data"\n  type: SLICE\n  bottom: "data"\n%s\n  slice_param { slice_dim: 1 }\n'
      '  include: { phase: TEST }\n}' % "\n".join('  top: "%s"' % x for x in d))
    a('layers {\n  name: "batch_concat_input_test"\n  type: CONCAT\n%s\n  top: "concat_input_datums"\n'
      '  concat_param { concat_dim: 0 }\n  include: { phase: TEST }\n}' % "\n".join('  bottom: "%s"' % x for x in d))
    a('layers {\n  name: "flatten_input"\n  type: FLATTEN\n  bottom: "concat_input_datums"\n'
      '  top: "concat_input_datums_flat"\n  include: { phase: TEST }\n}')
    f = ["test_sample_frame_%d" % j for j in range(1, frames + 1)]
    a('layers {\n  name: "slice_test"\n  type: SLICE\n  bottom: "concat_input_datums_flat"\n%s\n'
      '  slice_param { slice_dim: 0 }\n  include: { phase: TEST }\n}' % "\n".join('  top: "%s"' % x for x in f))
    a('layers {\n  name: "average_for_test"\n  type: ELTWISE\n%s\n  top: "original_feature"\n  eltwise_param {\n'
      '    operation: SUM\n%s\n  }\n  include: { phase: TEST }\n}'
      % ("\n".join('  bottom: "%s"' % x for x in f), "\n".join("    coeff: %.10g" % (1.0 / frames) for _ in f)))


def train_net(source, B, C, Nn, D, *, max_buffer=5000, swap=50, max_same=0, dropout=0.0, margin=2.0,
              norm="L2", name="videovec_train", w_std=0.001, test_source=None, test_batch=673, test_frames=4,
              id_to_class_file=None, id_to_weight_file=None, use_direct_weight=False, ip_regularization=0.0,
              context_type="WINDOW", rand_skip=0, negative_dataset=None, output_shot_distance=False,
              max_shot_distance=None):
    """test_source: also emit the TEST branch of the shipped file (window data -> average_for_test ->
    [shared fc7 / fc7_relu] -> test_norm -> retrieval_stats).
    id_to_weight_file / use_direct_weight: the weighted loss -- the data layer's video ids, replicated to (B, Nn)
    by a SUM layer, become MAX_MARGIN_LOSS's third bottom (max_margin_loss_layer.cpp:18-37, 82-97)."""
    weighted = bool(id_to_weight_file) or use_direct_weight
    L = []
    a = L.append
    a('name: "%s"' % name)
    if test_source:
        _test_branch_front(a, test_source, test_batch, test_frames)
    a('layers {\n  name: "shot_windows"\n  type: VIDEO_SAMPLED_SHOTS_DATA\n  top: "data"\n' +
      ('  top: "train_video_ids"\n' if weighted else '') +
      '  video_sampled_shots_data_param {\n    source: "%s"\n    backend: LMDB\n    batch_size: %d\n'
      '    num_negative_samples: %d\n    max_buffer_size: %d\n    negative_swap_percentage: %d\n'
      '    max_same_video_negs: %d\n    context_type: %s\n    context_size: %d\n%s  }\n'
      '  include: { phase: TRAIN }\n}' % (source, B, Nn, max_buffer, swap, max_same, context_type, C,
                                          ('    rand_skip: %d\n' % rand_skip if rand_skip else '') +
                                          ('    negative_dataset: "%s"\n' % negative_dataset if negative_dataset else '') +
                                          ('    output_shot_distance: true\n' if output_shot_distance else '') +
                                          ('    max_shot_distance: %g\n' % max_shot_distance if max_shot_distance is not None else '')))
    if context_type == "PAIRWISE":
        C = 2                    # video_sampled_shots_data_layer.cpp:200-201
    datums = ["target_datum"] + ["context_datum_%d" % j for j in range(1, C)] + \
             ["negative_datum_%d" % k for k in range(1, Nn + 1)]
    a('layers {\n  name: "slice_input_data"\n  type: SLICE\n  bottom: "data"\n%s\n  include: { phase: TRAIN }\n}'
      % "\n".join('  top: "%s"' % d for d in datums))
    a('layers {\n  name: "batch_concat_input"\n  type: CONCAT\n%s\n  top: "concat_input_datums"\n'
      '  concat_param { concat_dim: 0 }\n  include: { phase: TRAIN }\n}'
      % "\n".join('  bottom: "%s"' % d for d in datums))
    a('layers {\n  name: "flatten_input"\n  type: FLATTEN\n  bottom: "concat_input_datums"\n'
      '  top: "original_feature"\n  include: { phase: TRAIN }\n}')
    a('layers {\n  name: "fc7"\n  type: INNER_PRODUCT\n  bottom: "original_feature"\n  top: "ip1_nonorm"\n'
      '  blobs_lr: 1\n  blobs_lr: 2\n  weight_decay: 1\n  weight_decay: 0\n  inner_product_param {\n'
      '    num_output: %d\n%s    weight_filler { type: "gaussian" std: %g }\n'
      '    bias_filler { type: "constant" }\n  }\n}'
      % (D, "    regularization: %g\n" % ip_regularization if ip_regularization else "", w_std))
    a('layers {\n  name: "fc7_relu"\n  type: RELU\n  top: "ip2"\n  bottom: "ip1_nonorm"\n}')
    if dropout > 0:
        a('layers {\n  name: "drop2"\n  type: DROPOUT\n  bottom: "ip2"\n  top: "ip2"\n'
          '  dropout_param { dropout_ratio: %g }\n  include: { phase: TRAIN }\n}' % dropout)
    embs = ["target_emb_nonorm"] + ["context_window_emb_%d_nonorm" % j for j in range(1, C)] + \
           ["negative_emb_%d_nonorm" % k for k in range(1, Nn + 1)]
    a('layers {\n  name: "slice_emb"\n  type: SLICE\n  bottom: "ip2"\n%s\n  slice_param { slice_dim: 0 }\n'
      '  include: { phase: TRAIN }\n}' % "\n".join('  top: "%s"' % e for e in embs))
    if C > 2:
        a('layers {\n  name: "context_average"\n  type: ELTWISE\n%s\n  top: "context_feature_nonorm"\n'
          '  eltwise_param {\n    operation: SUM\n%s\n  }\n  include: { phase: TRAIN }\n}'
          % ("\n".join('  bottom: "%s"' % e for e in embs[1:C]),
             "\n".join("    coeff: %.10g" % (1.0 / (C - 1)) for _ in range(C - 1))))
    # one context frame: ELTWISE needs two bottoms (eltwise_layer.cpp:17), the embedding is normalised as it is
    a('layers {\n  name: "word_embedding_norm"\n  type: NORMALIZATION\n  bottom: "%s"\n'
      '  top: "context_feature"\n  include: { phase: TRAIN }\n}' % ("context_feature_nonorm" if C > 2 else embs[1]))
    pn = [embs[0]] + embs[C:]
    a('layers {\n  name: "concat_pos_neg_nonorm"\n  type: CONCAT\n  top: "pos_neg_nonorm"\n%s\n'
      '  concat_param { concat_dim: 0 }\n  include: { phase: TRAIN }\n}'
      % "\n".join('  bottom: "%s"' % e for e in pn))
    a('layers {\n  name: "pos_neg_norm"\n  type: NORMALIZATION\n  bottom: "pos_neg_nonorm"\n  top: "pos_neg"\n'
      '  include: { phase: TRAIN }\n}')
    normed = ["target_emb"] + ["negative_emb_%d" % k for k in range(1, Nn + 1)]
    a('layers {\n  name: "slice_pos_neg"\n  type: SLICE\n  bottom: "pos_neg"\n%s\n  slice_param { slice_dim: 0 }\n'
      '  include: { phase: TRAIN }\n}' % "\n".join('  top: "%s"' % e for e in normed))
    a('layers {\n  name: "prod_true"\n  type: ELTWISE\n  bottom: "context_feature"\n  bottom: "target_emb"\n'
      '  top: "target_prod"\n  eltwise_param { operation: PROD }\n  include: { phase: TRAIN }\n}')
    a('layers {\n  name: "sum_true"\n  type: SUM\n  bottom: "target_prod"\n  top: "target_score"\n'
      '  sum_param { num_output: %d }\n  include: { phase: TRAIN }\n}' % Nn)
    for k in range(1, Nn + 1):
        a('layers {\n  name: "prod_neg_%d"\n  type: ELTWISE\n  bottom: "context_feature"\n  bottom: "negative_emb_%d"\n'
          '  top: "neg_prod_%d"\n  eltwise_param { operation: PROD }\n  include: { phase: TRAIN }\n}' % (k, k, k))
        a('layers {\n  name: "sum_neg_%d"\n  type: SUM\n  bottom: "neg_prod_%d"\n  top: "neg_score_%d"\n'
          '  sum_param { num_output: 1 }\n  include: { phase: TRAIN }\n}' % (k, k, k))
    a('layers {\n  name: "concat_negative_scores"\n  type: CONCAT\n%s\n  top: "negative_scores"\n'
      '  concat_param { concat_dim: 1 }\n  include: { phase: TRAIN }\n}'
      % "\n".join('  bottom: "neg_score_%d"' % k for k in range(1, Nn + 1)))
    if weighted:
        a('layers {\n  name: "replicate_video_ids"\n  type: SUM\n  bottom: "train_video_ids"\n  top: "term_video_ids"\n'
          '  sum_param { num_output: %d }\n  include: { phase: TRAIN }\n}' % Nn)
    a('layers {\n  name: "max_margin_loss"\n  type: MAX_MARGIN_LOSS\n  bottom: "target_score"\n'
      '  bottom: "negative_scores"\n%s  top: "loss_output"\n  top: "train_violations"\n  loss_weight: 1.0\n'
      '  loss_weight: 0.0\n  max_margin_loss_param {\n    norm: %s\n    margin: %g\n%s%s  }\n'
      '  include: { phase: TRAIN }\n}'
      % ('  bottom: "term_video_ids"\n' if weighted else '', norm, margin,
         '    id_to_weight_file: "%s"\n' % id_to_weight_file if id_to_weight_file else '',
         '    use_direct_weight: true\n' if use_direct_weight else ''))
    if test_source:
        a('layers {\n  name: "test_norm"\n  type: NORMALIZATION\n  bottom: "ip2"\n  top: "ip2_norm"\n'
          '  include: { phase: TEST }\n}')
        a('layers {\n  name: "retrieval_stats"\n  type: RETRIEVAL_STATS\n  bottom: "ip2_norm"\n  bottom: "video_ids"\n'
          '  top: "test_map"\n  top: "test_hit_at_1"\n  top: "test_hit_at_5"\n  retrieval_stats_param {\n'
          '    id_to_class_file: "%s"\n  }\n  include: { phase: TEST }\n}' % id_to_class_file)
    return "\n".join(L) + "\n"


def extraction_net(source, batch, D, *, normalize=False, w_std=0.001):
    """The tail of projects/videovec_embedding/videovec_extraction.prototxt:179-205 (fc7 + ReLU `ip2`) on
    pre-extracted fc6 rows (one context frame per record stands for the upstream CaffeNet)."""
    L = ['name: "videovec_extraction"',
         'layers {\n  name: "fc6_rows"\n  type: VIDEO_SHOT_WINDOW_TEST_DATA\n  top: "fc6"\n  top: "label"\n'
         '  video_shot_window_test_data_param {\n    source: "%s"\n    backend: LMDB\n    batch_size: %d\n  }\n}'
         % (source, batch),
         'layers {\n  name: "fc7"\n  type: INNER_PRODUCT\n  bottom: "fc6"\n  top: "ip1_nonorm"\n  blobs_lr: 1\n'
         '  blobs_lr: 2\n  weight_decay: 1\n  weight_decay: 0\n  inner_product_param {\n    num_output: %d\n'
         '    weight_filler { type: "gaussian" std: %g }\n    bias_filler { type: "constant" }\n  }\n}' % (D, w_std),
         'layers {\n  name: "fc7_relu"\n  type: RELU\n  bottom: "ip1_nonorm"\n  top: "ip2"\n}']
    if normalize:
        L.append('layers {\n  name: "test_norm"\n  type: NORMALIZATION\n  bottom: "ip2"\n  top: "ip2_norm"\n}')
    return "\n".join(L) + "\n"


def solver(net_path, *, base_lr=0.001, momentum=0.9, weight_decay=0.0005, lr_policy="inv", gamma=0.001,
           power=0.75, stepsize=0, display=10, max_iter=100, snapshot=0, snapshot_prefix="videovec",
           random_seed=-1, snapshot_after_train=True, test_iter=0, test_interval=0, solver_type=None, delta=None,
           snapshot_diff=False):
    """Same fields as projects/videovec_embedding/mednet_embedding_train_solver.prototxt (+ solver_type / delta)."""
    s = ['net: "%s"' % net_path, "base_lr: %g" % base_lr, "momentum: %g" % momentum,
         "weight_decay: %g" % weight_decay, 'lr_policy: "%s"' % lr_policy, "gamma: %g" % gamma,
         "power: %g" % power, "display: %d" % display, "max_iter: %d" % max_iter,
         "snapshot: %d" % snapshot, 'snapshot_prefix: "%s"' % snapshot_prefix, "solver_mode: GPU"]
    if test_iter:
        s += ["test_iter: %d" % test_iter, "test_interval: %d" % test_interval]
    if stepsize:
        s.append("stepsize: %d" % stepsize)
    if random_seed >= 0:
        s.append("random_seed: %d" % random_seed)
    if not snapshot_after_train:
        s.append("snapshot_after_train: false")
    if solver_type:
        s.append("solver_type: %s" % solver_type)
    if delta is not None:
        s.append("delta: %g" % delta)
    if snapshot_diff:
        s.append("snapshot_diff: true")
    return "\n".join(s) + "\n"
