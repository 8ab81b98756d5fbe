"""Fused sampler kernels (nsa_sampler_sdf, nsa_sample_rays) vs the CPU oracle.  Needs an MI355X."""
import numpy as np
import pytest
import torch

from helpers import load, tt, params_of, oracle_config, draws_of, assert_close
from test_model_cpu import build_model
from test_oracle_golden import check_samples

pytestmark = pytest.mark.gpu


def _rays(fx, model):
    from nicer_slam_amd.utils.general import get_camera_from_tensor
    from nicer_slam_amd.utils import rend_util
    pose = get_camera_from_tensor(tt(fx["in_cam"]).cuda())
    d, o = rend_util.get_camera_params(tt(fx["in_uv"]).cuda(), pose, tt(fx["in_K"]).cuda())
    bs, n, _ = d.shape
    return d.reshape(-1, 3).contiguous(), o.unsqueeze(1).repeat(1, n, 1).reshape(-1, 3).contiguous()


@pytest.mark.parametrize("name", ["full_tracking", "full_tracking_poisson", "full_mapping", "full_vis_eval", "full_tracking_rw",
                                  "full_mapping_rw", "full_tracking_7scenes", "full_mapping_7scenes"])
def test_coarse_stage_sdf_and_z(name):
    """z (stratified) and coarse+fine SDF at the R*E coarse samples vs the oracle restatement."""
    from oracle import render_ref as R
    from nicer_slam_amd.fused import sampler as fs
    fx = load(name)
    model = build_model(fx).cuda()
    model.train(bool(fx["meta_training"]))
    model.voxels = tt(fx["in_voxels"]).cuda()
    assert fs.supported(model)
    d, o = _rays(fx, model)
    draws = draws_of(fx)
    t_rand = draws["t_rand"].cuda() if model.training else None
    z, sdf, far = fs.sampler_sdf(model, o, d, t_rand)
    cfg = oracle_config(fx)
    params = params_of(fx)
    zc, near_c, far_c = R.uniform_z(cfg, d.cpu(), o.cpu(), model.training, draws.get("t_rand"))
    assert_close(far, far_c.reshape(-1), 1e-6, 1e-6, "far")
    assert_close(z, zc, 1e-6, 1e-6, "z")
    pts = (o.cpu().unsqueeze(1) + zc.unsqueeze(2) * d.cpu().unsqueeze(1)).reshape(-1, 3)
    with torch.no_grad():
        sdf_c = R.sdf_vals(params, cfg, pts).reshape(zc.shape)
    assert_close(sdf, sdf_c, 1e-5, 1e-4, "sdf")


@pytest.mark.parametrize("name", ["full_tracking", "full_tracking_poisson", "full_mapping", "full_vis_eval", "full_tracking_rw",
                                  "full_mapping_rw", "full_tracking_7scenes", "full_mapping_7scenes"])
def test_full_sampler_vs_reference_samples(name):
    """End-to-end fused sampler vs the reference's own z_vals (goldens), compared in CDF space."""
    from oracle import render_ref as R
    from nicer_slam_amd.fused import sampler as fs
    fx = load(name)
    model = build_model(fx).cuda()
    model.train(bool(fx["meta_training"]))
    model.voxels = tt(fx["in_voxels"]).cuda()
    model.draws = draws_of(fx, "cuda")
    model.engine = "fused"
    d, o = _rays(fx, model)
    z_vals, z_eik = model.ray_sampler.get_z_vals(d, o, model)
    assert model.last_engine == "fused-sampler"
    # oracle CDF for the u-space comparison
    cfg, params = oracle_config(fx), params_of(fx)
    aux = {}
    zo, zo_eik = R.importance_z(params, cfg, d.cpu(), o.cpu(), tt(fx["in_voxels"]), model.training, draws_of(fx), aux=aux)
    # u-space tolerance 5e-5: a few float32 ulps of a cdf whose 1e-5-floor bins each carry ~1e-5 of mass
    # "_rw" cases: the cdf saturates after two bins, so the u = 1 sample (one of the 18 columns) lands on the last coarse
    # sample or just below it depending on whether cumsum's final value rounds to 1 or 1 + 2^-23 (searchsorted right=True,
    # ray_sampler.py:124-139) -- identical in CDF space, 4e-4 apart in z on every ray
    mt = 0.94 if name.endswith("_rw") else 0.97
    check_samples(z_vals.cpu(), tt(fx["out_z_vals"]), aux["bins"], aux["cdf"], u_tol=5e-5, min_tight=mt)
    check_samples(z_vals.cpu(), zo, aux["bins"], aux["cdf"], u_tol=5e-5, min_tight=mt)
    idx = draws_of(fx)["eik_idx"]
    assert_close(z_eik.cpu().reshape(-1), z_vals.cpu()[torch.arange(z_vals.shape[0]), idx], 0, 0, "z_eik")


def test_bench_shape_sampler_properties():
    """BASELINE size (1024 rays x 640 coarse + 128 final samples, shipped grids): sortedness, range, near/far present,
    and agreement of the SDF stage with the composed engine (torch ops + HIP hash operator) on the same points."""
    from nicer_slam_amd.model.network import SLAMNetwork
    from nicer_slam_amd.utils.conf import replica_model_conf
    from nicer_slam_amd.fused import sampler as fs

    class DS:
        img_res = (680, 1200)
    torch.manual_seed(0)
    conf = replica_model_conf(94, 640, 32, use_warp_loss=False)
    model = SLAMNetwork(conf, dataset=DS(), n_images=1,
                        colour_grid=dict(base_resolution=16, desired_resolution=64, log2_hashmap_size=12)).cuda()
    model.train()
    g = torch.Generator(device="cuda").manual_seed(3)
    for enc in (model.implicit_network.coarse.encoding, model.implicit_network.fine.encoding):
        enc.embeddings.data = (torch.rand(enc.embeddings.shape, device="cuda", generator=g) * 2 - 1) * 0.02
    R = 1024
    d = torch.nn.functional.normalize(torch.randn(R, 3, device="cuda", generator=g), dim=-1) * 0.7
    o = (torch.rand(R, 3, device="cuda", generator=g) - 0.5) * 0.4
    t_rand = torch.rand(R, 640, device="cuda", generator=g)
    z, sdf, far = fs.sampler_sdf(model, o, d, t_rand)
    pts = (o.unsqueeze(1) + z.unsqueeze(2) * d.unsqueeze(1)).reshape(-1, 3)
    with torch.no_grad():
        ref = model.implicit_network.get_sdf_vals(pts).reshape(z.shape)
    assert_close(sdf, ref, 2e-5, 1e-4, "sdf vs composed engine")
    model.engine = "fused"
    model.draws = {"t_rand": t_rand}
    z_vals, z_eik = model.ray_sampler.get_z_vals(d, o, model)
    assert z_vals.shape == (R, 128)
    assert bool((z_vals[:, 1:] >= z_vals[:, :-1]).all())
    assert float(z_vals.min()) == 0.0                      # near is always a sample
    assert bool((z_vals.max(dim=1)[0] >= far - 1e-6).all())   # far is always a sample
    assert bool(torch.isfinite(z_vals).all())
    # full-length rays (E = 640: ten samples per lane, the last lane holds the 1e10 interval) vs the oracle, CDF space
    from oracle import render_ref as Rr
    n = 48
    mk = Rr.make_grid_spec
    cfg = Rr.RenderConfig(coarse=Rr.SdfNetSpec(mk(4, 8, 32, 32, 19), 2), fine=Rr.SdfNetSpec(mk(8, 4, 32, 128, 19), 4),
                          colour_grid=mk(16, 2, 16, 64, 12), n_samples=94, n_samples_eval=640, n_samples_extra=32)
    params = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    extra = torch.arange(0, 640, 20)
    model.draws = {"t_rand": t_rand[:n], "extra_idx": extra.cuda(), "eik_idx": torch.zeros(n, dtype=torch.long).cuda()}
    z_gpu, _ = model.ray_sampler.get_z_vals(d[:n], o[:n], model)
    aux = {}
    z_cpu, _ = Rr.importance_z(params, cfg, d[:n].cpu(), o[:n].cpu(), torch.zeros(64, 64, 64), True,
                               {"t_rand": t_rand[:n].cpu(), "extra_idx": extra, "eik_idx": torch.zeros(n, dtype=torch.long)},
                               aux=aux)
    check_samples(z_gpu.cpu(), z_cpu, aux["bins"], aux["cdf"], u_tol=5e-5)


def test_draw_picks_is_a_permutation_prefix_and_uniform_ints():
    """nsa_draw_picks: extra_idx = indices of the n smallest keys (= prefix of the permutation that sorts the keys),
    eik_idx = floor(u*S); bit-exact integer work."""
    from nicer_slam_amd._native import lib, check
    torch.manual_seed(9)
    E, n_extra, R, S = 640, 32, 1000, 98
    u = torch.rand(E + R, device="cuda")
    u[5] = u[17]                                              # a tie: broken by index
    extra = torch.empty(n_extra, device="cuda", dtype=torch.int32)
    eik = torch.empty(R, device="cuda", dtype=torch.int32)
    check(lib.nsa_draw_picks(u.data_ptr(), E, n_extra, R, S, extra.data_ptr(), eik.data_ptr(),
                             torch.cuda.current_stream().cuda_stream))
    order = sorted(range(E), key=lambda i: (float(u[i]), i))[:n_extra]
    assert extra.tolist() == order
    assert eik.tolist() == [min(S - 1, int(float(v) * S)) for v in (u[E:] * 1.0).cpu()] or \
        torch.equal(eik.long().cpu(), (u[E:] * S).long().clamp(max=S - 1).cpu())
    assert int(eik.min()) >= 0 and int(eik.max()) <= S - 1 and len(set(eik.tolist())) > S // 2


def test_fast_draw_path_statistics():
    """Fused sampler with device-side draws (no pre-drawn tensors): valid sorted samples, extras distinct per call."""
    from nicer_slam_amd.utils.conf import replica_model_conf
    from nicer_slam_amd.model.network import SLAMNetwork
    from nicer_slam_amd.fused import sampler as fs
    torch.manual_seed(1)
    model = SLAMNetwork(replica_model_conf(use_warp_loss=False)).cuda().train()
    R = 256
    o = torch.tensor([0.1, 0.0, -0.2], device="cuda").repeat(R, 1)
    d = torch.nn.functional.normalize(torch.randn(R, 3, device="cuda"), dim=-1)
    z1, e1 = fs.get_z_vals(model, d, o)
    z2, e2 = fs.get_z_vals(model, d, o, need_eik=False)
    assert e2 is None and e1.shape == (R, 1)
    assert z1.shape == (R, 98) and bool((z1[:, 1:] >= z1[:, :-1]).all()) and bool(torch.isfinite(z1).all())
    assert not torch.equal(z1, z2)                            # fresh draws per call
    assert bool(((e1 >= z1[:, :1]) & (e1 <= z1[:, -1:])).all())


# ---- nsa_draw: the engine's own counter-based generator (Philox4x32-10) ---------------------------------------------------------
_M0, _M1, _W0, _W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85


def _philox4x32_10(ctr, key):
    """Pure-Python restatement of Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11);
    held to the Random123 known-answer vectors below."""
    c, k = list(ctr), list(key)
    for _ in range(10):
        p0, p1 = _M0 * c[0], _M1 * c[2]
        c = [(p1 >> 32) ^ c[1] ^ k[0], p1 & 0xFFFFFFFF, (p0 >> 32) ^ c[3] ^ k[1], p0 & 0xFFFFFFFF]
        k = [(k[0] + _W0) & 0xFFFFFFFF, (k[1] + _W1) & 0xFFFFFFFF]
    return c


def test_nsa_draw_is_philox_and_advances_its_own_state():
    import ctypes
    from nicer_slam_amd._native import lib, check
    assert _philox4x32_10([0] * 4, [0] * 2) == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]                   # Random123 kat_vectors
    assert _philox4x32_10([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2) == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    assert _philox4x32_10([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0]) == \
        [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]
    seed, call0 = 0x0123456789ABCDEF, (7 << 32) + 5
    R, E, n_extra, S = 37, 640, 32, 98
    n = R * E + 3                                            # ragged tail
    state = torch.tensor([seed, call0, 0, 0], dtype=torch.int64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    outs = []
    for rep in range(2):
        t = torch.full((n,), -1.0, device="cuda")
        ex = torch.full((n_extra,), -1, device="cuda", dtype=torch.int32)
        ek = torch.full((R,), -1, device="cuda", dtype=torch.int32)
        check(lib.nsa_draw(state.data_ptr(), n, t.data_ptr(), E, n_extra, R, S, ex.data_ptr(), ek.data_ptr(), st))
        torch.cuda.synchronize()
        outs.append((t.cpu(), ex.cpu(), ek.cpu()))
        assert state.cpu().tolist() == [seed, call0 + rep + 1, 0, 0]
    key = [seed & 0xFFFFFFFF, seed >> 32]
    for rep, (t, ex, ek) in enumerate(outs):
        call = call0 + rep
        ctr = lambda i, region: [i, region, call & 0xFFFFFFFF, call >> 32]
        for i in (0, 1, 2, 3, 4, 1023, 1024, 5000, n - 4, n - 1):
            want = (_philox4x32_10(ctr(i // 4, 0), key)[i % 4] >> 8) * 2.0 ** -24
            assert float(t[i]) == want, (rep, i)
        keys = [(_philox4x32_10(ctr(i // 4, 1), key)[i % 4] >> 8) * 2.0 ** -24 for i in range(E)]
        assert ex.tolist() == sorted(range(E), key=lambda i: (keys[i], i))[:n_extra]          # randperm(E)[:n] as ranks of E keys
        for r in (0, 1, 5, R - 1):
            u = (_philox4x32_10(ctr(r // 4, 2), key)[r % 4] >> 8) * 2.0 ** -24
            assert int(ek[r]) == min(int(np.float32(u) * np.float32(S)), S - 1)
        assert float(t.min()) >= 0.0 and float(t.max()) < 1.0 and abs(float(t.mean()) - 0.5) < 0.01
        assert abs(float(t.var()) - 1.0 / 12.0) < 0.005 and len(set(ex.tolist())) == n_extra
    assert not torch.equal(outs[0][0], outs[1][0]) and outs[0][1].tolist() != outs[1][1].tolist()
    # captured in a graph: every replay is a new call
    t = torch.empty(n, device="cuda")
    ex = torch.empty(n_extra, device="cuda", dtype=torch.int32)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        check(lib.nsa_draw(state.data_ptr(), n, t.data_ptr(), E, n_extra, 0, 0, ex.data_ptr(), None, side.cuda_stream))
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        check(lib.nsa_draw(state.data_ptr(), n, t.data_ptr(), E, n_extra, 0, 0, ex.data_ptr(), None,
                           torch.cuda.current_stream().cuda_stream))
    before = int(state[1])
    seen = []
    for _ in range(3):
        g.replay()
        torch.cuda.synchronize()
        seen.append(t.clone())
    assert int(state[1]) == before + 3 and not torch.equal(seen[0], seen[1]) and not torch.equal(seen[1], seen[2])
    # argument checks
    assert lib.nsa_draw(None, n, t.data_ptr(), E, n_extra, R, S, None, None, st) == 4
    assert lib.nsa_draw(state.data_ptr(), n, t.data_ptr(), 2048, n_extra, R, S, ex.data_ptr(), None, st) == 4


def test_free_running_sampler_uses_the_engines_generator_and_follows_the_seed():
    """Free-running fused sampler (no pinned draws): the draws come from nsa_draw, seeded through torch.manual_seed; two models
    seeded alike sample identically, successive calls differ, and the samples are as good as those drawn with torch.rand
    (same CDF-space statistics against the composed sampler is covered by the tests above; here: sorted, in range)."""
    from nicer_slam_amd.model.network import SLAMNetwork
    from nicer_slam_amd.utils.conf import replica_model_conf
    from nicer_slam_amd.fused import sampler as fs
    assert fs.OWN_RNG
    zs = []
    for rep in range(2):
        torch.manual_seed(77)
        model = SLAMNetwork(replica_model_conf(use_warp_loss=False), n_images=1,
                            colour_grid=dict(base_resolution=16, desired_resolution=64, log2_hashmap_size=12)).cuda().train()
        g = torch.Generator(device="cuda").manual_seed(3)
        d = torch.nn.functional.normalize(torch.randn(64, 3, device="cuda", generator=g), dim=-1)
        o = (torch.rand(64, 3, device="cuda", generator=g) - 0.5) * 0.4
        z1, e1 = fs.get_z_vals(model, d, o)
        z2, e2 = fs.get_z_vals(model, d, o)
        assert int(fs.draw_state(model)[1]) == 2
        assert not torch.equal(z1, z2)
        for z, e in ((z1, e1), (z2, e2)):
            assert bool((z[:, 1:] >= z[:, :-1]).all()) and bool(torch.isfinite(z).all()) and e.shape == (64, 1)
            assert bool(((e >= z[:, :1]) & (e <= z[:, -1:])).all())
        zs.append((z1, z2))
    assert torch.equal(zs[0][0], zs[1][0]) and torch.equal(zs[0][1], zs[1][1])
