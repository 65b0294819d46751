// net.cpp -- Net (reference: src/caffe/net.cpp) with the fused videovec plan.
#include <ctime>
#include "caffe/net.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace caffe {

namespace {
// what the graph matchers throw when the net is not (exactly) a videovec graph: Net::Init then runs it layer by layer
struct GraphMismatch { string what; };
}

template <typename Dtype>
Net<Dtype>::Net(const string& param_file, Caffe::Phase phase) {
  NetParameter param("NetParameter");
  pl::ReadProtoFromTextFileOrDie(param_file, &param);
  param.mutable_msg("state")->set_enum("phase", phase == Caffe::TRAIN ? "TRAIN" : "TEST");
  Init(param);
}

template <typename Dtype>
bool Net<Dtype>::StateMeetsRule(const NetState& state, const pl::Message& rule, const string& layer_name) {
  if (rule.has("phase") && rule.get_int("phase") != state.get_int("phase")) {
    LOG(INFO) << "The NetState phase (" << state.get_int("phase") << ") differed from the phase (" << rule.get_int("phase")
              << ") specified by a rule in layer " << layer_name;
    return false;
  }
  if (rule.has("min_level") && state.get_int("level") < rule.get_int("min_level")) return false;
  if (rule.has("max_level") && state.get_int("level") > rule.get_int("max_level")) return false;
  for (int i = 0; i < rule.size("stage"); ++i) {
    bool has = false;
    for (int j = 0; !has && j < state.size("stage"); ++j) has = rule.get_str("stage", i) == state.get_str("stage", j);
    if (!has) return false;
  }
  for (int i = 0; i < rule.size("not_stage"); ++i)
    for (int j = 0; j < state.size("stage"); ++j)
      if (rule.get_str("not_stage", i) == state.get_str("stage", j)) return false;
  return true;
}

template <typename Dtype>
void Net<Dtype>::FilterNet(const NetParameter& param, NetParameter* out) {
  NetState state(param.get_msg("state"));
  if (!state.has("phase")) state.set_enum("phase", Caffe::phase() == Caffe::TRAIN ? "TRAIN" : "TEST");
  *out = param;
  out->clear("layers");
  for (int i = 0; i < param.size("layers"); ++i) {
    const LayerParameter& lp = param.get_msg("layers", i);
    const string lname = lp.get_str("name");
    CHECK(lp.size("include") == 0 || lp.size("exclude") == 0) << "Specify either include rules or exclude rules; not both.";
    bool included = lp.size("include") == 0;
    for (int j = 0; included && j < lp.size("exclude"); ++j) if (StateMeetsRule(state, lp.get_msg("exclude", j), lname)) included = false;
    for (int j = 0; !included && j < lp.size("include"); ++j) if (StateMeetsRule(state, lp.get_msg("include", j), lname)) included = true;
    if (included) *out->add_msg("layers") = lp;
  }
}

template <typename Dtype>
void Net<Dtype>::Init(const NetParameter& in_param) {
  NetParameter param("NetParameter");
  FilterNet(in_param, &param);                                                        // net.cpp:36-37
  LOG(INFO) << "Initializing net from parameters: " << param.size("layers") << " layers after phase filtering";
  name_ = param.get_str("name");
  CHECK_EQ(param.size("input"), 0) << "net-level input blobs are not used by the videovec graph";
  std::set<string> available, consumed;
  const int L = param.size("layers");
  bottom_vecs_.resize(L); top_vecs_.resize(L); bottom_id_vecs_.resize(L); top_id_vecs_.resize(L);
  for (int li = 0; li < L; ++li) {
    const LayerParameter& lp = param.get_msg("layers", li);
    layers_.push_back(shared_ptr<Layer<Dtype> >(GetLayer<Dtype>(lp)));                // net.cpp:63-66
    layer_names_.push_back(lp.get_str("name"));
    layer_names_index_[lp.get_str("name")] = li;
    LOG(INFO) << "Creating Layer " << lp.get_str("name");
    for (int b = 0; b < lp.size("bottom"); ++b) {                                     // AppendBottom, net.cpp:381-402
      const string bn = lp.get_str("bottom", b);
      CHECK(available.count(bn)) << "Unknown blob input " << bn << " (at index " << b << ") to layer " << li;
      const int id = blob_names_index_[bn];
      LOG(INFO) << lp.get_str("name") << " <- " << bn;
      bottom_vecs_[li].push_back(blobs_[id].get());
      bottom_id_vecs_[li].push_back(id);
      consumed.insert(bn);     // the reference erases here and relies on InsertSplits for fan-out
                               // (insert_splits.cpp); the fused plan sums fan-in diffs itself, so no
                               // SPLIT layers are materialised and a blob may simply be read again
    }
    for (int t = 0; t < lp.size("top"); ++t) {                                        // AppendTop, net.cpp:333-379
      const string tn = lp.get_str("top", t);
      if (t < lp.size("bottom") && lp.get_str("bottom", t) == tn) {                   // in-place
        LOG(INFO) << lp.get_str("name") << " -> " << tn << " (in-place)";
        const int id = blob_names_index_[tn];
        top_vecs_[li].push_back(blobs_[id].get());
        top_id_vecs_[li].push_back(id);
      } else {
        CHECK(!blob_names_index_.count(tn)) << "Duplicate blobs produced by multiple sources.";
        LOG(INFO) << lp.get_str("name") << " -> " << tn;
        blobs_.push_back(shared_ptr<Blob<Dtype> >(new Blob<Dtype>()));
        blob_names_.push_back(tn);
        blob_names_index_[tn] = (int)blobs_.size() - 1;
        top_vecs_[li].push_back(blobs_.back().get());
        top_id_vecs_[li].push_back((int)blobs_.size() - 1);
      }
      available.insert(tn);
    }
    layers_[li]->SetUp(bottom_vecs_[li], &top_vecs_[li]);                            // net.cpp:97
    blob_loss_weights_.resize(blobs_.size(), Dtype(0));
    for (size_t t = 0; t < top_vecs_[li].size(); ++t) {
      blob_loss_weights_[top_id_vecs_[li][t]] = layers_[li]->loss((int)t);
      LOG(INFO) << "Top shape: " << top_vecs_[li][t]->num() << " " << top_vecs_[li][t]->channels() << " "
                << top_vecs_[li][t]->height() << " " << top_vecs_[li][t]->width() << " (" << top_vecs_[li][t]->count() << ")";
      if (layers_[li]->loss((int)t)) LOG(INFO) << "    with loss weight " << layers_[li]->loss((int)t);
    }
    // parameter bookkeeping (net.cpp:99-140, 404-464): lr / decay multipliers per blob
    const int nb = (int)layers_[li]->blobs().size();
    CHECK(lp.size("blobs_lr") == nb || lp.size("blobs_lr") == 0) << "Incorrect blobs lr size: should be either 0 or the same as the number of the layer's parameter blobs.";
    CHECK(lp.size("weight_decay") == nb || lp.size("weight_decay") == 0) << "Incorrect blobs weight decay size";
    CHECK_EQ(lp.size("param"), 0) << "shared parameters (param:) are not used by the videovec graph";
    for (int b = 0; b < nb; ++b) {
      params_.push_back(layers_[li]->blobs()[b]);
      params_lr_.push_back(lp.size("blobs_lr") ? (float)lp.get_num("blobs_lr", b) : 1.f);
      params_weight_decay_.push_back(lp.size("weight_decay") ? (float)lp.get_num("weight_decay", b) : 1.f);
    }
  }
  // remaining blobs are the net outputs (net.cpp:200-208; a std::set, hence sorted by name)
  for (const string& n : available) {
    if (consumed.count(n)) continue;
    LOG(INFO) << "This network produces output " << n;
    net_output_blobs_.push_back(blobs_[blob_names_index_[n]].get());
    net_output_blob_indices_.push_back(blob_names_index_[n]);
  }
  bool is_test = false;
  for (size_t li = 0; li < layers_.size(); ++li) is_test |= layers_[li]->type() == "VIDEO_SHOT_WINDOW_TEST_DATA";
  // The recognised videovec graphs run as ONE fused plan; anything else built from these layer classes runs layer by
  // layer (ForwardFromTo / BackwardFromTo over Layer::Forward_gpu / Backward_gpu).  VV_FACADE_SEQUENTIAL=1 forces the
  // layer-by-layer executor on a graph the matcher would accept (tests: both executors must agree).
  sequential_ = getenv("VV_FACADE_SEQUENTIAL") && atoi(getenv("VV_FACADE_SEQUENTIAL")) != 0;
  if (!sequential_) {
    try {
      if (is_test) MatchVideovecTestGraph(); else MatchVideovecTrainGraph();
    } catch (const GraphMismatch& m) {
      LOG(INFO) << "Not the fused videovec pattern -- " << m.what << ".  Running this net layer by layer.";
      sequential_ = true;
    }
  }
  if (sequential_) SetUpSequential(is_test);
  if (is_test) {
    // a TEST / extraction net has its own feature table, hence its own context on the same device
    VV_CHECK(vv_create(Caffe::device(), getenv("VV_PREC") && !strcmp(getenv("VV_PREC"), "bf16") ? VV_PREC_BF16 : VV_PREC_F16, &ctx_));
    own_ctx_ = true;
    Caffe::register_ctx(ctx_);
    static_cast<VideoShotWindowTestDataLayer<Dtype>*>(layers_[plan_.data_layer].get())->dataset()->UploadTable(ctx_);
  } else {
    ctx_ = Caffe::ctx();
    static_cast<VideoSampledShotsDataLayer<Dtype>*>(layers_[plan_.data_layer].get())->dataset()->UploadTable(ctx_);
  }
  PushParamsToDevice();
  if (!is_test && Caffe::world() > 1) {
    // data-parallel TRAIN net: the gradient buffer is summed over the ranks before every update (vv_apply_update joins
    // the all-reduce); VV_COMM=shm selects the one-device test transport; the update overlaps the next forward GEMM
    // (vv_comm_overlap: F-chunks on the communication stream, gated forward) unless VV_COMM_OVERLAP=0
    const string id_path = "/tmp/vv_caffe_comm_" + Caffe::job_id();
    const bool shm = getenv("VV_COMM") && !strcmp(getenv("VV_COMM"), "shm");
    const bool peer = getenv("VV_COMM") && !strcmp(getenv("VV_COMM"), "peer");   // the one-shot direct exchange over peer mappings
    VV_CHECK(vv_comm_init(ctx_, Caffe::world(), Caffe::rank(), id_path.c_str(), shm ? VV_COMM_SHM : peer ? VV_COMM_PEER : VV_COMM_RCCL));
    VV_CHECK(vv_comm_overlap(ctx_, !(getenv("VV_COMM_OVERLAP") && atoi(getenv("VV_COMM_OVERLAP")) == 0)));
    // VV_COMM_SCHEDULE=sync | overlap | sharded names the schedule outright (sharded: reduce-scatter, the update on this rank's rows,
    // all-gather of the 16-bit copy: include/videovec.h, vv_comm_schedule)
    if (const char* sch = getenv("VV_COMM_SCHEDULE")) {
      const int which = !strcmp(sch, "sync") ? 0 : !strcmp(sch, "overlap") ? 1 : !strcmp(sch, "sharded") ? 2 : -1;
      CHECK_GE(which, 0) << "VV_COMM_SCHEDULE=" << sch << ": sync, overlap or sharded";
      VV_CHECK(vv_comm_schedule(ctx_, which));
    }
    cfg_.global_count = (int64_t)Caffe::world() * plan_.B * plan_.Nn;
    LOG(INFO) << "Data-parallel rank " << Caffe::rank() << " of " << Caffe::world() << ": per-GPU batch " << plan_.B
              << ", global batch " << Caffe::world() * plan_.B << ", gradients all-reduced over "
              << (shm ? "shared memory (test transport)" : peer ? "direct peer mappings (one-shot reduce-scatter / all-gather)" : "RCCL");
  }
  LOG(INFO) << "Network initialization done.";
}

// ---------------------------------------------------------------------------------------------
// Graph matcher: symbolic forward execution of the filtered graph.  Every blob gets a description
// of what it is in terms of the fused plan; a layer whose inputs fit no rule is fatal.
// ---------------------------------------------------------------------------------------------
namespace {
enum Kind { K_NONE, K_DATA, K_DATUM, K_XROWS, K_Y, K_H, K_EMB, K_CTXMEAN, K_CTXNORM, K_PN, K_PNNORM, K_PNORM,
            K_PROD, K_SCORE, K_NEGSCORES, K_LOSS, K_VIOL, K_LABEL, K_LABELREP };
struct Sym { Kind k = K_NONE; int a = 0; int reps = 1; };
}

template <typename Dtype>
void Net<Dtype>::MatchVideovecTrainGraph() {
  const char* why = "This build runs only the videovec_embedding TRAIN graph "
                    "(projects/videovec_embedding/mednet_embedding_train.prototxt); ";
  std::map<int, Sym> sym;      // blob id -> symbol
  FusedPlan& P = plan_;
  int CN = 0;
  bool have_dropout = false;
  for (size_t li = 0; li < layers_.size(); ++li) {
    Layer<Dtype>* layer = layers_[li].get();
    const string type = layer->type();
    const string lname = layer_names_[li];
    const vector<int>& bi = bottom_id_vecs_[li];
    const vector<int>& ti = top_id_vecs_[li];
    auto in = [&](int i) -> Sym { return sym[bi[i]]; };
    auto bad = [&](const string& what) { throw GraphMismatch{string(why) + "layer " + lname + " (" + type + "): " + what}; };
    if (type == "VIDEO_SAMPLED_SHOTS_DATA") {
      if (P.data_layer >= 0) bad("second data layer");
      auto* d = static_cast<VideoSampledShotsDataLayer<Dtype>*>(layer);
      P.data_layer = (int)li; P.B = d->batch_size(); P.C = d->context_size(); P.Nn = d->num_negative_samples(); P.F = d->feature_size();
      CN = P.C + P.Nn;
      if (P.Nn < 1) bad("num_negative_samples must be >= 1 for the ranking loss");
      sym[ti[0]].k = K_DATA;
      if (ti.size() > 1) sym[ti[1]].k = K_LABEL;
    } else if (type == "SLICE") {
      const int dim = (int)layer->layer_param().get_msg("slice_param").get_int("slice_dim");
      const Sym s = in(0);
      if (s.k == K_DATA && dim == 1) {
        if ((int)ti.size() != CN) bad("the input slice must produce context_size + num_negative_samples tops");
        for (int c = 0; c < CN; ++c) { sym[ti[c]].k = K_DATUM; sym[ti[c]].a = c; }
      } else if (s.k == K_H && dim == 0) {
        if ((int)ti.size() != CN) bad("the embedding slice must produce context_size + num_negative_samples tops");
        for (int c = 0; c < CN; ++c) { sym[ti[c]].k = K_EMB; sym[ti[c]].a = c; }
      } else if (s.k == K_PNNORM && dim == 0) {
        if ((int)ti.size() != 1 + P.Nn) bad("the normalised target/negative slice must produce 1 + num_negative_samples tops");
        for (int q = 0; q <= P.Nn; ++q) { sym[ti[q]].k = K_PNORM; sym[ti[q]].a = q; }
      } else bad("unexpected SLICE input");
    } else if (type == "CONCAT") {
      const int dim = (int)layer->layer_param().get_msg("concat_param").get_int("concat_dim");
      if (in(0).k == K_DATUM && dim == 0) {
        if ((int)bi.size() != CN) bad("the input concat must take every sliced datum");
        for (int c = 0; c < CN; ++c) if (in(c).k != K_DATUM || in(c).a != c) bad("datum order must be target, context.., negatives.. (channel-major rows)");
        sym[ti[0]].k = K_XROWS;
      } else if (in(0).k == K_EMB && dim == 0) {
        if ((int)bi.size() != 1 + P.Nn) bad("concat_pos_neg must take the target and every negative embedding");
        if (in(0).a != 0) bad("first bottom must be the target embedding");
        for (int q = 1; q <= P.Nn; ++q) if (in(q).k != K_EMB || in(q).a != P.C + q - 1) bad("negative embeddings out of order");
        sym[ti[0]].k = K_PN;
      } else if (in(0).k == K_SCORE && dim == 1) {
        if ((int)bi.size() != P.Nn) bad("the score concat must take every negative score");
        for (int q = 0; q < P.Nn; ++q) if (in(q).k != K_SCORE || in(q).a != q + 1 || in(q).reps != 1) bad("negative scores out of order");
        sym[ti[0]].k = K_NEGSCORES;
        P.negative_scores_blob = blob_names_[ti[0]];
      } else bad("unexpected CONCAT inputs");
    } else if (type == "FLATTEN") {
      if (in(0).k != K_XROWS) bad("FLATTEN is expected on the concatenated input rows");
      sym[ti[0]].k = K_XROWS;
    } else if (type == "INNER_PRODUCT") {
      if (in(0).k != K_XROWS) bad("fc layer must consume the gathered feature rows");
      if (P.ip_layer >= 0) bad("second INNER_PRODUCT layer");
      P.ip_layer = (int)li;
      P.D = static_cast<InnerProductLayer<Dtype>*>(layer)->num_output();
      if (!layer->layer_param().get_msg("inner_product_param").get_bool("bias_term")) bad("bias_term: false is not built");
      P.ip_regularization = (float)layer->layer_param().get_msg("inner_product_param").get_num("regularization");
      sym[ti[0]].k = K_Y;
    } else if (type == "RELU") {
      if (in(0).k != K_Y) bad("RELU is expected on the fc output");
      if (layer->layer_param().get_msg("relu_param").get_num("negative_slope") != 0) bad("negative_slope != 0 is not built");
      sym[ti[0]].k = K_H;
      P.ip2_blob = blob_names_[ti[0]];
    } else if (type == "DROPOUT") {
      if (in(0).k != K_H || have_dropout) bad("DROPOUT is expected once, on the ReLU output");
      have_dropout = true;
      P.dropout_ratio = (float)layer->layer_param().get_msg("dropout_param").get_num("dropout_ratio");
      sym[ti[0]].k = K_H;
      P.ip2_blob = blob_names_[ti[0]];
    } else if (type == "ELTWISE") {
      auto* e = static_cast<EltwiseLayer<Dtype>*>(layer);
      if (e->op() == "SUM") {
        if ((int)bi.size() != P.C - 1) bad("context_average must sum the C-1 context embeddings");
        for (int j = 0; j < P.C - 1; ++j) if (in(j).k != K_EMB || in(j).a != j + 1) bad("context embeddings out of order");
        P.ctx_coeff.assign(e->coeffs().begin(), e->coeffs().end());
        sym[ti[0]].k = K_CTXMEAN;
      } else if (e->op() == "PROD") {
        if (bi.size() != 2) bad("PROD takes the context feature and one embedding");
        if (!layer->layer_param().get_msg("eltwise_param").get_bool("stable_prod_grad")) bad("stable_prod_grad: false is not built");
        Sym a = in(0), b = in(1);
        if (a.k == K_PNORM && b.k == K_CTXNORM) std::swap(a, b);
        if (a.k != K_CTXNORM || b.k != K_PNORM) bad("PROD must pair context_feature with a normalised target / negative embedding");
        sym[ti[0]].k = K_PROD; sym[ti[0]].a = b.a;
      } else bad("Eltwise MAX is not part of the graph");
    } else if (type == "NORMALIZATION") {
      if (in(0).k == K_CTXMEAN) sym[ti[0]].k = K_CTXNORM;
      else if (in(0).k == K_EMB && in(0).a == 1 && P.C == 2) {     // one context frame (PAIRWISE): nothing to average
        P.ctx_coeff.assign(1, 1.f);
        sym[ti[0]].k = K_CTXNORM;
      }
      else if (in(0).k == K_PN) sym[ti[0]].k = K_PNNORM;
      else bad("unexpected NORMALIZATION input");
    } else if (type == "SUM" && in(0).k == K_LABEL) {
      // video ids replicated over the Nn loss terms of an item: the weighted loss's third bottom
      if ((int)layer->layer_param().get_msg("sum_param").get_num("num_output") != P.Nn) bad("the replicated video ids need num_output == num_negative_samples");
      sym[ti[0]].k = K_LABELREP;
    } else if (type == "SUM") {
      if (in(0).k != K_PROD) bad("SUM is expected on a PROD output");
      const int reps = (int)layer->layer_param().get_msg("sum_param").get_num("num_output");
      sym[ti[0]].k = K_SCORE; sym[ti[0]].a = in(0).a; sym[ti[0]].reps = reps;
      if (in(0).a == 0) {
        if (reps != P.Nn) bad("sum_true num_output must equal num_negative_samples (quirk Q3)");
        P.target_score_blob = blob_names_[ti[0]];
      } else if (reps != 1) bad("negative score SUM layers must have num_output 1");
    } else if (type == "MAX_MARGIN_LOSS") {
      if (P.loss_layer >= 0) bad("second loss layer");
      if (bi.size() == 3) {
        if (in(2).k != K_LABELREP && !(in(2).k == K_LABEL && P.Nn == 1)) bad("the third bottom must be the data layer's video ids replicated to (B, Nn) by a SUM layer");
        P.weighted_loss = true;
      }
      if (in(0).k != K_SCORE || in(0).a != 0 || in(1).k != K_NEGSCORES) bad("bottoms must be target_score, negative_scores");
      P.loss_layer = (int)li;
      const pl::Message& mp = layer->layer_param().get_msg("max_margin_loss_param");
      P.margin = (float)mp.get_num("margin");
      P.norm = mp.get_enum("norm") == "L2" ? VV_NORM_L2 : VV_NORM_L1;
      P.loss_weight = (float)layer->loss(0);
      if (ti.size() > 1 && layer->loss(1) != 0) bad("a loss weight on train_violations is not built");
      sym[ti[0]].k = K_LOSS; P.loss_blob = blob_names_[ti[0]];
      if (ti.size() > 1) { sym[ti[1]].k = K_VIOL; P.violations_blob = blob_names_[ti[1]]; }
    } else if (type == "SPLIT") {
      for (size_t t = 0; t < ti.size(); ++t) sym[ti[t]] = in(0);
    } else {
      bad("layer type outside the path");
    }
  }
  if (P.data_layer < 0 || P.ip_layer < 0 || P.loss_layer < 0)
    throw GraphMismatch{string(why) + "the graph needs a VIDEO_SAMPLED_SHOTS_DATA layer, one INNER_PRODUCT layer and a MAX_MARGIN_LOSS layer"};
  blob_sym_.assign(blobs_.size(), BlobSym());
  for (auto& kv : sym) { blob_sym_[kv.first].kind = kv.second.k; blob_sym_[kv.first].a = kv.second.a; blob_sym_[kv.first].reps = kv.second.reps; }
  vv_step_cfg_default(&cfg_);
  cfg_.B = P.B; cfg_.C = P.C; cfg_.Nn = P.Nn;
  cfg_.margin = P.margin; cfg_.norm = P.norm; cfg_.loss_weight = P.loss_weight;
  cfg_.ctx_coeff = P.ctx_coeff.data();
  cfg_.dropout_ratio = P.dropout_ratio;
  cfg_.ip_regularization = P.ip_regularization;
  cfg_.dropout_seed = Caffe::random_seed();
  // blobs_lr / weight_decay multipliers of fc7's {W, b} (mednet_embedding_train.prototxt:195-198)
  int p0 = 0;
  for (int li = 0; li < P.ip_layer; ++li) p0 += (int)layers_[li]->blobs().size();
  for (int k = 0; k < 2; ++k) { cfg_.lr_mult[k] = params_lr_[p0 + k]; cfg_.decay_mult[k] = params_weight_decay_[p0 + k]; }
  LOG(INFO) << "Fused videovec plan: B=" << P.B << " C=" << P.C << " Nn=" << P.Nn << " F=" << P.F << " D=" << P.D
            << " margin=" << P.margin << " norm=L" << P.norm << " dropout=" << P.dropout_ratio;
}

// ---------------------------------------------------------------------------------------------
// Layer-by-layer execution (Net::ForwardFromTo / BackwardFromTo, net.cpp:501-514, 567-578) for nets that are not the
// fused pattern.  The one INNER_PRODUCT layer still owns the context's parameters, gradient buffer and update kernel,
// so Solver / Update / snapshots work unchanged; every other blob lives in its SyncedMemory.
// Fan-out: the reference rewrites the net with SPLIT layers (insert_splits.cpp) so that the diffs of a blob read by
// several layers are summed.  Here no layer is inserted (blob names stay those of the prototxt): during the backward
// sweep the first consumer of a blob writes its diff, every later one adds to it (Layer::Backward, accumulate_bottom).
// ---------------------------------------------------------------------------------------------
template <typename Dtype>
void Net<Dtype>::SetUpSequential(bool is_test) {
  plan_ = FusedPlan();
  blob_sym_.clear();
  plan_.test = is_test;
  const char* data_type = is_test ? "VIDEO_SHOT_WINDOW_TEST_DATA" : "VIDEO_SAMPLED_SHOTS_DATA";
  for (size_t li = 0; li < layers_.size(); ++li) {
    const string type = layers_[li]->type();
    if (type == data_type) { CHECK_LT(plan_.data_layer, 0) << "one data layer per net"; plan_.data_layer = (int)li; }
    if (type == "INNER_PRODUCT") {
      CHECK_LT(plan_.ip_layer, 0) << "the context holds the parameters of ONE fc layer (the videovec path has one)";
      plan_.ip_layer = (int)li;
      plan_.D = static_cast<InnerProductLayer<Dtype>*>(layers_[li].get())->num_output();
      plan_.F = bottom_vecs_[li][0]->count() / bottom_vecs_[li][0]->num();
    }
    if (type == "RETRIEVAL_STATS") plan_.stats_layer = (int)li;
  }
  CHECK_GE(plan_.data_layer, 0) << "the net needs a " << data_type << " layer";
  CHECK_GE(plan_.ip_layer, 0) << "the net needs an INNER_PRODUCT layer";
  if (is_test) {
    auto* d = static_cast<VideoShotWindowTestDataLayer<Dtype>*>(layers_[plan_.data_layer].get());
    plan_.B = d->batch_size(); plan_.C = d->context_size();
    CHECK_EQ(plan_.F, d->dataset()->F) << "the fc layer must read rows of the data layer's feature size";
  } else {
    auto* d = static_cast<VideoSampledShotsDataLayer<Dtype>*>(layers_[plan_.data_layer].get());
    plan_.B = d->batch_size(); plan_.C = d->context_size(); plan_.Nn = d->num_negative_samples();
    CHECK_EQ(plan_.F, d->feature_size()) << "the fc layer must read rows of the data layer's feature size";
  }
  vv_step_cfg_default(&cfg_);
  cfg_.B = plan_.B; cfg_.C = plan_.C; cfg_.Nn = plan_.Nn > 0 ? plan_.Nn : 1;
  int p0 = 0;
  for (int li = 0; li < plan_.ip_layer; ++li) p0 += (int)layers_[li]->blobs().size();
  for (int k = 0; k < 2; ++k) { cfg_.lr_mult[k] = params_lr_[p0 + k]; cfg_.decay_mult[k] = params_weight_decay_[p0 + k]; }
  // which blobs and layers take part in the backward sweep (net.cpp:99-140): a layer needs it when a bottom does or
  // when it owns parameters with a nonzero learning rate
  vector<bool> blob_bw(blobs_.size(), false);
  layer_need_backward_.assign(layers_.size(), false);
  bottom_need_backward_.assign(layers_.size(), vector<bool>());
  for (size_t li = 0; li < layers_.size(); ++li) {
    bool need = false;
    for (int id : bottom_id_vecs_[li]) { bottom_need_backward_[li].push_back(blob_bw[id]); need = need || blob_bw[id]; }
    if ((int)li == plan_.ip_layer) for (int k = 0; k < 2; ++k) need = need || cfg_.lr_mult[k] != 0.f;
    layer_need_backward_[li] = need && !is_test;
    for (int id : top_id_vecs_[li]) blob_bw[id] = blob_bw[id] || layer_need_backward_[li];
  }
  LOG(INFO) << "Layer-by-layer plan: " << layers_.size() << " layers, B=" << plan_.B << " F=" << plan_.F << " D=" << plan_.D;
}

template <typename Dtype>
Dtype Net<Dtype>::ForwardFromTo(int start, int end) {
  Dtype loss = 0;
  for (int li = start; li <= end; ++li) {
    layers_[li]->Reshape(bottom_vecs_[li], &top_vecs_[li]);                             // net.cpp:508 (quirk Q11)
    const Dtype l = layers_[li]->Forward(bottom_vecs_[li], &top_vecs_[li]);
    loss += l;
    if (debug_info_) LOG(INFO) << "    [Forward] Layer " << layer_names_[li] << ", loss " << l;
  }
  return loss;
}

template <typename Dtype>
void Net<Dtype>::BackwardFromTo(int start, int end) {
  vector<bool> written(blobs_.size(), false);
  for (int li = start; li >= end; --li) {
    if (!layer_need_backward_[li]) continue;
    vector<bool> acc(bottom_id_vecs_[li].size(), false);
    for (size_t i = 0; i < bottom_id_vecs_[li].size(); ++i) {
      const int id = bottom_id_vecs_[li][i];
      const bool in_place = i < top_id_vecs_[li].size() && top_id_vecs_[li][i] == id;
      if (!bottom_need_backward_[li][i] || in_place) continue;
      acc[i] = written[id];
      written[id] = true;
    }
    layers_[li]->set_accumulate_bottom(acc);
    layers_[li]->Backward(top_vecs_[li], bottom_need_backward_[li], &bottom_vecs_[li]);
  }
}

// The net's blobs and layers free their device memory through the context they were allocated under: release them
// while a net-owned context still exists.
template <typename Dtype>
Net<Dtype>::~Net() {
  if (own_ctx_ && ctx_) {
    net_output_blobs_.clear();
    bottom_vecs_.clear(); top_vecs_.clear();
    blobs_.clear(); params_.clear(); layers_.clear();
    Caffe::unregister_ctx(ctx_);
    vv_destroy(ctx_);
  }
}

namespace {
// the per-layer operators run on whatever context is current; a TEST net owns a second one
struct CurrentCtx {
  vv_ctx* saved;
  explicit CurrentCtx(vv_ctx* c) : saved(Caffe::set_current_ctx(c)) {}
  ~CurrentCtx() { Caffe::set_current_ctx(saved); }
};
}

template <typename Dtype>
Dtype Net<Dtype>::SequentialStep() {
  CurrentCtx guard(ctx_);
  const int last = (int)layers_.size() - 1;
  const Dtype loss = ForwardFromTo(0, last);
  if (!plan_.test) BackwardFromTo(last, 0);
  ++iter_;
  last_loss_ = loss;
  return loss;
}

template <typename Dtype>
void Net<Dtype>::PushParamsToDevice() {
  Layer<Dtype>* ip = layers_[plan_.ip_layer].get();
  CHECK_EQ(ip->blobs()[0]->count(), plan_.D * plan_.F);
  VV_CHECK(vv_params_set(ctx_, plan_.D, ip->blobs()[0]->cpu_data(), ip->blobs()[1]->cpu_data(), NULL, NULL));
  params_stale_ = false;
}
template <typename Dtype>
void Net<Dtype>::PullParamsFromDevice() {
  if (!params_stale_) return;
  Layer<Dtype>* ip = layers_[plan_.ip_layer].get();
  VV_CHECK(vv_params_get(ctx_, ip->blobs()[0]->mutable_cpu_data(), ip->blobs()[1]->mutable_cpu_data(), NULL, NULL));
  params_stale_ = false;
}
template <typename Dtype>
vector<shared_ptr<Blob<Dtype> > >& Net<Dtype>::params() { PullParamsFromDevice(); return params_; }

template <typename Dtype>
void Net<Dtype>::GetHistory(vector<shared_ptr<Blob<Dtype> > >* history) {
  history->resize(2);
  (*history)[0].reset(new Blob<Dtype>(1, 1, plan_.D, plan_.F));
  (*history)[1].reset(new Blob<Dtype>(1, 1, 1, plan_.D));
  VV_CHECK(vv_params_get(ctx_, NULL, NULL, (*history)[0]->mutable_cpu_data(), (*history)[1]->mutable_cpu_data()));
}
template <typename Dtype>
void Net<Dtype>::SetHistory(const vector<shared_ptr<Blob<Dtype> > >& history) {
  CHECK_EQ((int)history.size(), 2) << "Incorrect length of history blobs.";
  CHECK_EQ(history[0]->count(), plan_.D * plan_.F); CHECK_EQ(history[1]->count(), plan_.D);
  PullParamsFromDevice();
  Layer<Dtype>* ip = layers_[plan_.ip_layer].get();
  VV_CHECK(vv_params_set(ctx_, plan_.D, ip->blobs()[0]->cpu_data(), ip->blobs()[1]->cpu_data(),
                         history[0]->cpu_data(), history[1]->cpu_data()));
}

namespace {
// VV_FACADE_PROFILE=1: where the host thread of `caffe train` spends an iteration (printed at exit)
struct HostProf {
  double t_batch = 0, t_fb = 0, t_loss = 0; long n = 0; bool on = getenv("VV_FACADE_PROFILE") != nullptr;
  static double now() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + ts.tv_nsec * 1e-9; }
  ~HostProf() {
    if (on && n) fprintf(stderr, "facade host profile over %ld iterations: wait-for-batch %.1f us, forward_backward call %.1f us, "
                                 "loss read-back %.1f us (averages)\n", n, t_batch / n * 1e6, t_fb / n * 1e6, t_loss / n * 1e6);
  }
} g_hprof;
}

template <typename Dtype>
Dtype Net<Dtype>::ForwardBackward(const vector<Blob<Dtype>*>&) {
  if (sequential_) return SequentialStep();
  if (plan_.test) return ForwardTest();
  auto* data = static_cast<VideoSampledShotsDataLayer<Dtype>*>(layers_[plan_.data_layer].get());
  const double tp0 = g_hprof.on ? HostProf::now() : 0;
  data->NextBatch(&idx_, &last_src_, &label_);
  const double tp1 = g_hprof.on ? HostProf::now() : 0;
  cfg_.ctx_coeff = plan_.ctx_coeff.data();
  cfg_.item_weight = nullptr;
  if (plan_.weighted_loss) {
    auto* ml = static_cast<MaxMarginLossLayer<Dtype>*>(layers_[plan_.loss_layer].get());
    item_weight_.resize(plan_.B);
    for (int i = 0; i < plan_.B; ++i) item_weight_[i] = ml->WeightOf((float)label_[i]);
    cfg_.item_weight = item_weight_.data();
  }
  bool q1 = false;
  for (size_t i = 0; i < idx_.size() && !q1; ++i) q1 = idx_[i] != last_src_[i];
  if (q1) VV_CHECK(vv_forward_backward_q1(ctx_, &cfg_, idx_.data(), last_src_.data()));
  else VV_CHECK(vv_forward_backward(ctx_, &cfg_, idx_.data(), 0));
  ++iter_;
  const double tp2 = g_hprof.on ? HostProf::now() : 0;
  if (g_hprof.on) { g_hprof.t_batch += tp1 - tp0; g_hprof.t_fb += tp2 - tp1; ++g_hprof.n; }
  if (!loss_needed_) return last_loss_;
  float loss = 0, viol = 0;
  VV_CHECK(vv_loss_get(ctx_, &loss, &viol));
  if (g_hprof.on) g_hprof.t_loss += HostProf::now() - tp2;
  // keep the two scalar output blobs current (they feed the solver's display lines)
  if (has_blob(plan_.loss_blob)) blobs_[blob_names_index_[plan_.loss_blob]]->mutable_cpu_data()[0] = loss / (plan_.loss_weight ? plan_.loss_weight : 1.f);
  if (!plan_.violations_blob.empty()) blobs_[blob_names_index_[plan_.violations_blob]]->mutable_cpu_data()[0] = viol;
  last_loss_ = loss;
  return loss;
}
template <typename Dtype>
const vector<Blob<Dtype>*>& Net<Dtype>::Forward(const vector<Blob<Dtype>*>& bottom, Dtype* loss) {
  const Dtype l = ForwardBackward(bottom);
  if (loss) *loss = l;
  return net_output_blobs_;
}
template <typename Dtype>
const vector<Blob<Dtype>*>& Net<Dtype>::ForwardPrefilled(Dtype* loss) { return Forward(vector<Blob<Dtype>*>(), loss); }

template <typename Dtype>
void Net<Dtype>::SetUpdateHyperParams(float rate, float momentum, float weight_decay, const string& reg, int solver_type,
                                      float delta) {
  cfg_.lr = rate; cfg_.momentum = momentum; cfg_.weight_decay = weight_decay;
  cfg_.solver_type = solver_type; cfg_.delta = delta;
  if (reg == "L2") cfg_.reg = VV_REG_L2;
  else if (reg == "L1") cfg_.reg = VV_REG_L1;
  else LOG(FATAL) << "Unknown regularization type: " << reg;                          // solver.cpp:523
}
template <typename Dtype>
void Net<Dtype>::ShareTrainedLayersWith(Net* other) {
  CHECK_EQ(plan_.D, other->plan_.D); CHECK_EQ(plan_.F, other->plan_.F);
  other->PullParamsFromDevice();
  Layer<Dtype>* src = other->layers_[other->plan_.ip_layer].get();
  Layer<Dtype>* dst = layers_[plan_.ip_layer].get();
  CHECK(layer_names_[plan_.ip_layer] == other->layer_names_[other->plan_.ip_layer])
      << "layers are shared by name (net.cpp:641-648)";
  for (int k = 0; k < 2; ++k) dst->blobs()[k]->CopyFrom(*src->blobs()[k]);
  PushParamsToDevice();
}

// ---------------------------------------------------------------------------------------------
// TEST / extraction graph (mednet_embedding_train.prototxt TEST phase: 30-45, 75-104, 133-177, 190-217,
// 344-352, 673-688; videovec_extraction.prototxt:179-205): VIDEO_SHOT_WINDOW_TEST_DATA -> SLICE dim 1
// -> CONCAT dim 0 -> FLATTEN -> SLICE dim 0 -> ELTWISE SUM(coeff) -> fc -> RELU [-> NORMALIZATION]
// [-> RETRIEVAL_STATS].  With a single context frame the data blob may feed FLATTEN / fc directly.
// ---------------------------------------------------------------------------------------------
template <typename Dtype>
void Net<Dtype>::MatchVideovecTestGraph() {
  const char* why = "This build runs only the TEST / extraction graph of projects/videovec_embedding; ";
  enum TK { T_NONE, T_DATA, T_DATUM, T_XROWS, T_FRAME, T_MEAN, T_Y, T_H, T_HN, T_LABEL, T_STAT };
  struct TS { TK k = T_NONE; int a = 0; };
  std::map<int, TS> sym;
  FusedPlan& P = plan_;
  P.test = true;
  int k = 0;
  for (size_t li = 0; li < layers_.size(); ++li) {
    Layer<Dtype>* layer = layers_[li].get();
    const string type = layer->type(), lname = layer_names_[li];
    const vector<int>& bi = bottom_id_vecs_[li];
    const vector<int>& ti = top_id_vecs_[li];
    auto in = [&](int i) -> TS { return sym[bi[i]]; };
    auto bad = [&](const string& what) { throw GraphMismatch{string(why) + "layer " + lname + " (" + type + "): " + what}; };
    if (type == "VIDEO_SHOT_WINDOW_TEST_DATA") {
      auto* d = static_cast<VideoShotWindowTestDataLayer<Dtype>*>(layer);
      P.data_layer = (int)li; P.B = d->batch_size(); P.C = k = d->context_size(); P.F = d->dataset()->F;
      sym[ti[0]].k = k == 1 ? T_MEAN : T_DATA;
      if (k == 1) P.ctx_coeff.assign(1, 1.f);
      if (ti.size() > 1) { sym[ti[1]].k = T_LABEL; P.label_blob = blob_names_[ti[1]]; }
    } else if (type == "SLICE") {
      const int dim = (int)layer->layer_param().get_msg("slice_param").get_int("slice_dim");
      if (in(0).k == T_DATA && dim == 1 && (int)ti.size() == k) for (int j = 0; j < k; ++j) { sym[ti[j]].k = T_DATUM; sym[ti[j]].a = j; }
      else if (in(0).k == T_XROWS && dim == 0 && (int)ti.size() == k) for (int j = 0; j < k; ++j) { sym[ti[j]].k = T_FRAME; sym[ti[j]].a = j; }
      else bad("unexpected SLICE");
    } else if (type == "CONCAT") {
      const int dim = (int)layer->layer_param().get_msg("concat_param").get_int("concat_dim");
      if (dim != 0 || (int)bi.size() != k) bad("unexpected CONCAT");
      for (int j = 0; j < k; ++j) if (in(j).k != T_DATUM || in(j).a != j) bad("context datums out of order");
      sym[ti[0]].k = T_XROWS;
    } else if (type == "FLATTEN") {
      if (in(0).k != T_XROWS && in(0).k != T_MEAN) bad("unexpected FLATTEN input");
      sym[ti[0]] = in(0);
    } else if (type == "ELTWISE") {
      auto* e = static_cast<EltwiseLayer<Dtype>*>(layer);
      if (e->op() != "SUM" || (int)bi.size() != k) bad("average_for_test must SUM the context frames");
      for (int j = 0; j < k; ++j) if (in(j).k != T_FRAME || in(j).a != j) bad("frames out of order");
      P.ctx_coeff.assign(e->coeffs().begin(), e->coeffs().end());
      sym[ti[0]].k = T_MEAN;
    } else if (type == "INNER_PRODUCT") {
      if (in(0).k != T_MEAN || P.ip_layer >= 0) bad("the fc layer must consume the averaged input");
      P.ip_layer = (int)li; P.D = static_cast<InnerProductLayer<Dtype>*>(layer)->num_output();
      sym[ti[0]].k = T_Y; P.ip1_blob = blob_names_[ti[0]]; P.ip2_blob = P.ip1_blob;
    } else if (type == "RELU") {
      if (in(0).k != T_Y) bad("RELU is expected on the fc output");
      P.relu = true; sym[ti[0]].k = T_H; P.ip2_blob = blob_names_[ti[0]];
    } else if (type == "NORMALIZATION") {
      if (in(0).k != T_H && in(0).k != T_Y) bad("unexpected NORMALIZATION input");
      sym[ti[0]].k = T_HN; P.norm_blob = blob_names_[ti[0]];
    } else if (type == "RETRIEVAL_STATS") {
      if (in(1).k != T_LABEL) bad("second bottom must be the video ids");
      if (in(0).k != T_HN && in(0).k != T_H) bad("first bottom must be the (normalised) embedding");
      if (in(0).k == T_H) P.norm_blob.clear();
      P.stats_layer = (int)li;
      for (int t = 0; t < 3; ++t) { sym[ti[t]].k = T_STAT; P.stat_blobs[t] = blob_names_[ti[t]]; }
    } else bad("layer type outside the TEST path");
  }
  if (P.data_layer < 0 || P.ip_layer < 0) throw GraphMismatch{string(why) + "need a VIDEO_SHOT_WINDOW_TEST_DATA layer and one INNER_PRODUCT layer"};
  LOG(INFO) << "Fused videovec TEST plan: B=" << P.B << " frames=" << P.C << " F=" << P.F << " D=" << P.D
            << (P.stats_layer >= 0 ? " + retrieval stats" : "");
}

template <typename Dtype>
Dtype Net<Dtype>::ForwardTest() {
  auto* data = static_cast<VideoShotWindowTestDataLayer<Dtype>*>(layers_[plan_.data_layer].get());
  data->NextBatch(&idx_, &label_);
  Blob<Dtype>* ip2 = blobs_[blob_names_index_[plan_.ip2_blob]].get();
  CHECK_EQ(ip2->count(), plan_.B * plan_.D);
  VV_CHECK(vv_embed_mean(ctx_, idx_.data(), plan_.B, plan_.C, plan_.ctx_coeff.data(), plan_.relu ? 1 : 0, 0,
                         ip2->mutable_cpu_data()));
  if (!plan_.label_blob.empty()) {
    Dtype* l = blobs_[blob_names_index_[plan_.label_blob]]->mutable_cpu_data();
    for (int i = 0; i < plan_.B; ++i) l[i] = (Dtype)label_[i];
  }
  const Dtype* feat = ip2->cpu_data();
  if (!plan_.norm_blob.empty()) {      // NORMALIZATION forward (normalization_layer.cpp:29-61): y = x / (|x| + 1e-10)
    Blob<Dtype>* nb = blobs_[blob_names_index_[plan_.norm_blob]].get();
    for (int i = 0; i < plan_.B; ++i) {
      const Dtype* x = ip2->cpu_data() + (size_t)i * plan_.D;
      double s = 0;
      for (int d = 0; d < plan_.D; ++d) s += (double)x[d] * x[d];
      const Dtype inv = (Dtype)(1.0 / (std::sqrt(s) + 1e-10));
      Dtype* y = nb->mutable_cpu_data() + (size_t)i * plan_.D;
      for (int d = 0; d < plan_.D; ++d) y[d] = x[d] * inv;
    }
    feat = nb->cpu_data();
  }
  if (plan_.stats_layer >= 0) {
    auto* rs = static_cast<RetrievalStatsLayer<Dtype>*>(layers_[plan_.stats_layer].get());
    float m = 0, h1 = 0, h5 = 0;
    VV_CHECK(vv_retrieval_stats(ctx_, feat, plan_.B, plan_.D, label_.data(), rs->map_ids().data(), rs->map_cls().data(),
                                (int)rs->map_ids().size(), rs->exclude_same_video_shots() ? 1 : 0, &m, &h1, &h5));
    const float v[3] = {m, h1, h5};
    for (int t = 0; t < 3; ++t) blobs_[blob_names_index_[plan_.stat_blobs[t]]]->mutable_cpu_data()[0] = v[t];
  }
  ++iter_;
  return 0;
}

template <typename Dtype>
void Net<Dtype>::HintUpdate() {
  if (sequential_ || plan_.test || debug_info_ || !ctx_) return;
  VV_CHECK(vv_update_hint(ctx_, &cfg_));
}
template <typename Dtype>
void Net<Dtype>::Update() {
  VV_CHECK(vv_apply_update(ctx_, &cfg_));
  params_stale_ = true;
}

template <typename Dtype>
bool Net<Dtype>::has_blob(const string& n) { return blob_names_index_.count(n) != 0; }
template <typename Dtype>
bool Net<Dtype>::has_layer(const string& n) { return layer_names_index_.count(n) != 0; }
template <typename Dtype>
const shared_ptr<Layer<Dtype> > Net<Dtype>::layer_by_name(const string& n) {
  if (!has_layer(n)) { LOG(WARNING) << "Unknown layer name " << n; return shared_ptr<Layer<Dtype> >(); }
  return layers_[layer_names_index_[n]];
}
template <typename Dtype>
const shared_ptr<Blob<Dtype> > Net<Dtype>::blob_by_name(const string& n) {
  if (!has_blob(n)) { LOG(WARNING) << "Unknown blob name " << n; return shared_ptr<Blob<Dtype> >(); }   // net.cpp:846-857
  shared_ptr<Blob<Dtype> > b = blobs_[blob_names_index_[n]];
  if (iter_ > 0 && !plan_.test && !sequential_) MaterializeTrainBlob(blob_names_index_[n]);
  return b;
}

// The fused plan keeps only ip2, the scores and the gradients on the device.  Any other named blob of the TRAIN graph
// is rebuilt here, on demand, from those and from the feature table, following the layer that produces it in the
// reference (slice_layer.cpp, concat_layer.cpp, eltwise_layer.cpp:52-73, normalization_layer.cpp:29-48, sum_layer.cpp:31-54):
// a debugging / inspection path (Net::blob_by_name, net.cpp:846-857), not a hot one.
template <typename Dtype>
void Net<Dtype>::MaterializeTrainBlob(int id) {
  if (blob_sym_.empty()) return;
  const BlobSym sy = blob_sym_[id];
  Blob<Dtype>* b = blobs_[id].get();
  const int B = plan_.B, C = plan_.C, Nn = plan_.Nn, CN = C + Nn, D = plan_.D, F = plan_.F;
  Dtype* out = b->mutable_cpu_data();
  auto feature_rows = [&](const vector<int32_t>& rows, const vector<int32_t>& lasts, Dtype* dst) {
    // rows of the table; -1 = all-zero; a slot whose last feature comes from another row (quirk Q1) is patched
    vector<int32_t> valid; vector<size_t> where;
    for (size_t i = 0; i < rows.size(); ++i) if (rows[i] >= 0) { valid.push_back(rows[i]); where.push_back(i); }
    std::fill(dst, dst + rows.size() * (size_t)F, Dtype(0));
    if (valid.empty()) return;
    vector<float> tmp(valid.size() * (size_t)F);
    VV_CHECK(vv_table_get(ctx_, valid.data(), (int64_t)valid.size(), tmp.data()));
    for (size_t k = 0; k < valid.size(); ++k) memcpy(dst + where[k] * (size_t)F, &tmp[k * (size_t)F], sizeof(float) * F);
    for (size_t i = 0; i < rows.size(); ++i) {
      if (lasts[i] == rows[i]) continue;
      float lastv = 0.f;
      if (lasts[i] >= 0) { vector<float> r1((size_t)F); int32_t rr = lasts[i]; VV_CHECK(vv_table_get(ctx_, &rr, 1, r1.data())); lastv = r1[F - 1]; }
      dst[i * (size_t)F + F - 1] = lastv;
    }
  };
  auto slot = [&](int bb, int ch) { return (size_t)bb * CN + ch; };
  if (sy.kind == K_DATA) {                                     // (B, C+Nn, F, 1), item-major
    feature_rows(idx_, last_src_, out);
    return;
  }
  if (sy.kind == K_DATUM || sy.kind == K_XROWS) {              // one channel (B,1,F,1) / all channels channel-major (CN*B, F)
    vector<int32_t> rows, lasts;
    for (int ch = sy.kind == K_DATUM ? sy.a : 0; ch < (sy.kind == K_DATUM ? sy.a + 1 : CN); ++ch)
      for (int bb = 0; bb < B; ++bb) { rows.push_back(idx_[slot(bb, ch)]); lasts.push_back(last_src_[slot(bb, ch)]); }
    feature_rows(rows, lasts, out);
    return;
  }
  if (sy.kind == K_LABEL) { for (int i = 0; i < B; ++i) out[i] = (Dtype)label_[i]; return; }
  if (sy.kind == K_LABELREP) { for (int i = 0; i < B; ++i) for (int k = 0; k < Nn; ++k) out[(size_t)i * Nn + k] = (Dtype)label_[i]; return; }
  if (sy.kind == K_LOSS || sy.kind == K_VIOL || sy.kind == K_NONE) return;      // scalars are kept current by ForwardBackward
  if (sy.kind == K_SCORE && sy.a == 0) { CHECK_EQ(b->count(), B * Nn); VV_CHECK(vv_blobs_get(ctx_, NULL, out, NULL, NULL)); return; }
  if (sy.kind == K_NEGSCORES) { CHECK_EQ(b->count(), B * Nn); VV_CHECK(vv_blobs_get(ctx_, NULL, NULL, out, NULL)); return; }
  if (sy.kind == K_SCORE) {                                    // neg_score_q = column q-1 of negative_scores
    vector<float> ns((size_t)B * Nn);
    VV_CHECK(vv_blobs_get(ctx_, NULL, NULL, ns.data(), NULL));
    for (int i = 0; i < B; ++i) out[i] = ns[(size_t)i * Nn + sy.a - 1];
    return;
  }
  if (sy.kind == K_Y) {                                        // ip1_nonorm: the projection without ReLU, recomputed
    CHECK(last_src_ == idx_) << "ip1_nonorm of a batch with quirk-Q1 slots is not rebuilt";
    vector<int32_t> rows;
    for (int ch = 0; ch < CN; ++ch) for (int bb = 0; bb < B; ++bb) rows.push_back(idx_[slot(bb, ch)]);
    for (int32_t r : rows) CHECK_GE(r, 0) << "ip1_nonorm of a batch with empty slots is not rebuilt";
    VV_CHECK(vv_embed(ctx_, rows.data(), (int64_t)rows.size(), 0, 0, out));
    return;
  }
  // everything else is a function of ip2 (channel-major rows ch*B + b)
  vector<float> H((size_t)CN * B * D);
  VV_CHECK(vv_blobs_get(ctx_, H.data(), NULL, NULL, NULL));
  auto hrow = [&](int ch, int bb) { return &H[((size_t)ch * B + bb) * D]; };
  auto normalize = [&](const float* x, float* y) {            // normalization_layer.cpp:29-48
    double s = 0; for (int d = 0; d < D; ++d) s += (double)x[d] * x[d];
    const float nrm = (float)std::sqrt(s) + 1e-10f;
    for (int d = 0; d < D; ++d) y[d] = x[d] / nrm;
  };
  auto ctx_mean = [&](int bb, float* y) {
    for (int d = 0; d < D; ++d) { float s = 0; for (int j = 1; j < C; ++j) s += plan_.ctx_coeff[j - 1] * hrow(j, bb)[d]; y[d] = s; }
  };
  vector<float> t1(D), t2(D);
  switch (sy.kind) {
    case K_H: memcpy(out, H.data(), sizeof(float) * H.size()); break;
    case K_EMB: memcpy(out, hrow(sy.a, 0), sizeof(float) * (size_t)B * D); break;
    case K_CTXMEAN: for (int bb = 0; bb < B; ++bb) ctx_mean(bb, out + (size_t)bb * D); break;
    case K_CTXNORM: for (int bb = 0; bb < B; ++bb) { ctx_mean(bb, t1.data()); normalize(t1.data(), out + (size_t)bb * D); } break;
    case K_PN: case K_PNNORM:
      for (int q = 0; q <= Nn; ++q) for (int bb = 0; bb < B; ++bb) {
        const float* x = hrow(q == 0 ? 0 : C + q - 1, bb);
        Dtype* y = out + ((size_t)q * B + bb) * D;
        if (sy.kind == K_PN) memcpy(y, x, sizeof(float) * D); else normalize(x, y);
      }
      break;
    case K_PNORM: for (int bb = 0; bb < B; ++bb) normalize(hrow(sy.a == 0 ? 0 : C + sy.a - 1, bb), out + (size_t)bb * D); break;
    case K_PROD:
      for (int bb = 0; bb < B; ++bb) {
        ctx_mean(bb, t1.data()); normalize(t1.data(), t2.data());
        normalize(hrow(sy.a == 0 ? 0 : C + sy.a - 1, bb), t1.data());
        for (int d = 0; d < D; ++d) out[(size_t)bb * D + d] = t2[d] * t1[d];
      }
      break;
    default: break;
  }
}

template <typename Dtype>
void Net<Dtype>::CopyTrainedLayersFrom(const NetParameter& param) {
  // the host blobs must hold the trained parameters before any of them is overwritten: a source model without the fc
  // layer (or one ignored by name) leaves it untouched, as the reference does (net.cpp:703-706)
  PullParamsFromDevice();
  for (int i = 0; i < param.size("layers"); ++i) {
    const LayerParameter& src = param.get_msg("layers", i);
    const string sname = src.get_str("name");
    if (!layer_names_index_.count(sname)) { LOG(INFO) << "Ignoring source layer " << sname; continue; }   // net.cpp:703-706
    LOG(INFO) << "Copying source layer " << sname;
    vector<shared_ptr<Blob<Dtype> > >& tgt = layers_[layer_names_index_[sname]]->blobs();
    CHECK_EQ((int)tgt.size(), src.size("blobs")) << "Incompatible number of blobs for layer " << sname;
    for (size_t j = 0; j < tgt.size(); ++j) {
      const pl::Message& bp = src.get_msg("blobs", (int)j);
      CHECK_EQ(tgt[j]->num(), bp.get_int("num")); CHECK_EQ(tgt[j]->channels(), bp.get_int("channels"));
      CHECK_EQ(tgt[j]->height(), bp.get_int("height")); CHECK_EQ(tgt[j]->width(), bp.get_int("width"));
      tgt[j]->FromProto(bp);
    }
  }
  // keep the momentum history that is on the device
  vector<shared_ptr<Blob<Dtype> > > hist;
  GetHistory(&hist);
  params_stale_ = false;
  Layer<Dtype>* ip = layers_[plan_.ip_layer].get();
  VV_CHECK(vv_params_set(ctx_, plan_.D, ip->blobs()[0]->cpu_data(), ip->blobs()[1]->cpu_data(),
                         hist[0]->cpu_data(), hist[1]->cpu_data()));
}
template <typename Dtype>
void Net<Dtype>::CopyTrainedLayersFrom(const string trained_filename) {
  NetParameter param("NetParameter");
  pl::ReadProtoFromBinaryFileOrDie(trained_filename, &param);
  CopyTrainedLayersFrom(param);
}
template <typename Dtype>
void Net<Dtype>::ToProto(NetParameter* param, bool write_diff) {
  PullParamsFromDevice();
  if (write_diff) {
    Layer<Dtype>* ip = layers_[plan_.ip_layer].get();
    if (iter_ > 0) VV_CHECK(vv_grads_get(ctx_, ip->blobs()[0]->mutable_cpu_diff(), ip->blobs()[1]->mutable_cpu_diff()));
  }
  *param = NetParameter("NetParameter");
  param->set_str("name", name_);
  LOG(INFO) << "Serializing " << layers_.size() << " layers";                          // net.cpp:784
  for (size_t i = 0; i < layers_.size(); ++i) {
    LayerParameter* lp = param->add_msg("layers");
    layers_[i]->ToProto(lp, write_diff);
  }
}

template class Net<float>;

}  // namespace caffe
