"""vv_box_probe: the calibration record of a bench line -- two fixed probes of the library on buffers of their own."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_box_probe_reports_plausible_rates_and_leaves_the_engine_alone():
    import videovector_amd as vv
    from videovector_amd.synth import SyntheticVideos, init_weights
    B, C, Nn, F, D = 32, 5, 2, 128, 32
    ds = SyntheticVideos(seed=1701, n_videos=50)
    idx = vv.Sampler(ds.video_id, ds.n_shots, ds.row_base, batch_size=B, context_size=C, num_negative_samples=Nn,
                     max_buffer_size=500, negative_swap_percentage=50).next()
    W, b = init_weights(1, D, F, std=0.02)
    eng = vv.Engine(0, "f16")
    eng.table_synth(ds.seed, ds.n_rows, F)
    eng.params_set(W, b)
    cfg = vv.StepConfig(B, C, Nn, lr=0.01)
    eng.forward_backward(cfg, idx)
    loss0 = eng.loss()
    p = eng.box_probe()
    # MI355X: the gathered-free forward instantiation runs at 0.9-1.4 PFLOP/s, the chip holds 1.3-2.4 GHz under it, a copy streams at 3-7 TB/s
    assert 300.0 < p["gemm_tflops"] < 2500.0, p
    assert 1000.0 < p["gemm_clock_mhz"] < 2500.0, p
    assert 1.0 < p["copy_tbs"] < 8.0, p
    assert p["gemm_shape"] == [20736, 4096, 512] and p["gemm_launches"] == 24 and p["copy_bytes"] == 1 << 30
    assert abs(p["gemm_tflops"] - 2.0 * 20736 * 4096 * 512 / (p["gemm_ms"] * 1e-3) / 1e12) < 1e-6 * p["gemm_tflops"]
    q = eng.box_probe()                                       # repeatable within the box's own wobble
    assert abs(q["gemm_tflops"] - p["gemm_tflops"]) < 0.15 * p["gemm_tflops"]
    eng.forward_backward(cfg, idx)                            # the engine's own state is untouched
    assert eng.loss() == loss0
    assert np.isfinite(eng.grads()[0]).all()
