"""CPU-side tests: C-ABI library loads and exports every declared symbol (no compute without a GPU),
host logic (config, metrics, timers, synthetic data, failure without a GPU), and the 2-rank
aggregation used by bench.py (gloo)."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_abi_library_loads_and_exports_declared_symbols():
    from deformationpyramid_amd import _native
    path = _native.build()
    L = ctypes.CDLL(path)
    header = open(os.path.join(ROOT, "include", "ndp_hip.h")).read()
    declared = set(re.findall(r"^(?:int|const char \*)\s*\*?(ndp_[a-z0-9_]+)\s*\(", header, flags=re.M))
    assert declared, "no declarations found in include/ndp_hip.h"
    assert declared == set(_native.EXPORTS), declared ^ set(_native.EXPORTS)
    for name in declared:
        assert getattr(L, name) is not None
    L.ndp_version.restype = ctypes.c_int
    assert L.ndp_version() >= 100


def test_struct_layouts_match_the_header():
    from deformationpyramid_amd import _native
    assert ctypes.sizeof(_native.PairState) == 120           # 12 ints/floats, 1 double, 16 ints
    assert _native.PairState.loss_prev.offset == 48
    assert ctypes.sizeof(_native.PairGeom) == 16
    assert _native.Engine.break_threshold_ratio.offset % 8 == 0
    assert _native.Engine.geom.offset % 8 == 0
    assert ctypes.sizeof(_native.WarpJob) == 48 and ctypes.sizeof(_native.LoadJob) == 88     # 5 / 8 pointers + 2 / 6 ints
    header = open(os.path.join(ROOT, "include", "ndp_hip.h")).read()
    assert int(re.search(r"#define NDP_MAX_WARP_JOBS (\d+)", header).group(1)) == _native.MAX_WARP_JOBS
    assert int(re.search(r"#define NDP_MAX_LOAD_JOBS (\d+)", header).group(1)) == _native.MAX_LOAD_JOBS


def test_the_package_reads_no_environment_variable_and_variants_are_explicit():
    """Nothing under deformationpyramid_amd/ reads the environment (the one WRITE is GPU_MAX_HW_QUEUES' default, before HIP starts): an
    experiment build of the library can only be selected by _native.use_variant() -- which the measurement tools call for
    NDP_HIP_LIB -- and not once the product library is loaded; bench.py refuses a selected variant without --allow-variant."""
    import glob
    from deformationpyramid_amd import _native
    pkg = os.path.join(ROOT, "deformationpyramid_amd")
    for path in glob.glob(os.path.join(pkg, "**", "*"), recursive=True):
        if not path.endswith((".py", ".cpp", ".hip", ".inc", ".h")):
            continue
        text = open(path).read()
        assert "getenv" not in text and "environ.get" not in text and "environ[" not in text, path
    _native.lib()
    with pytest.raises(_native.NdpError):
        _native.use_variant("/nonexistent/libndp_variant.so")
    assert _native.variant() is None
    bench = open(os.path.join(ROOT, "bench.py")).read()
    assert "N.variant() and not args.allow_variant" in bench and "tools" not in [ln for ln in bench.splitlines() if ln.startswith(("import", "from"))]


def test_invalid_arguments_are_rejected_without_a_gpu():
    from deformationpyramid_amd import _native
    from deformationpyramid_amd.layout import LayerDesc
    L = _native.lib()
    bad = LayerDesc(width=257).c_struct()                            # (1..256 x depth 1..4 are served since round 6: csrc/ndp_generic.inc)
    rc = L.ndp_level_fwd(ctypes.byref(bad), None, 0, -8, None, 0, None, None, None, None, None)
    assert rc == -2 and b"width must be 1..256" in L.ndp_last_error()
    bad = LayerDesc(width=64, n_hidden=4).c_struct()
    rc = L.ndp_level_fwd(ctypes.byref(bad), None, 0, -8, None, 0, None, None, None, None, None)
    assert rc == -2 and b"depth 1..4" in L.ndp_last_error()
    ok = LayerDesc().c_struct()
    assert L.ndp_level_fwd(ctypes.byref(ok), None, 0, -8, None, 5, None, None, None, None, None) == -1
    assert L.ndp_chamfer_nn_fwd(None, 0, None, 0, None, None, None, None, None) == -1
    assert L.ndp_pair_means(None, 0, None, 0, None, None) == -1
    assert L.ndp_pyramid_fwd_batch(ctypes.byref(ok), 9, -8, 8, None, 1, None) == -1            # p_stride < P
    assert L.ndp_engine_load(None, 0, None, 0, None) == -1
    deep = LayerDesc(n_hidden=4).c_struct()                     # depth 5: beyond what the generic kernels (and the oracle) carry
    assert L.ndp_level_fwd(ctypes.byref(deep), None, 0, -8, None, 0, None, None, None, None, None) == -2
    for fmt in ("axis_angle", "euler", "quaternion", "6D"):     # n = 0 is a valid no-op for every served variant
        for gate in (False, True):
            d = LayerDesc(rotfmt=fmt, nonrigidity=gate).c_struct()
            assert L.ndp_level_fwd(ctypes.byref(d), ctypes.c_void_p(16), 0, -8, None, 0, None, None, None, None, None) == 0


def test_no_cpu_fallback():
    from deformationpyramid_amd import ops, _native
    from deformationpyramid_amd.config import Config
    from deformationpyramid_amd.registration import Registration
    with pytest.raises(_native.NdpError):
        ops.chamfer_nn(torch.zeros(4, 3), torch.zeros(4, 3))
    cfg = Config(deformation_model="NDP", device=torch.device("cpu"), depth=3, width=128, k0=-8, m=2, w_reg=0.0,
                 rotation_format="axis_angle", motion_type="SE3", samples=10, iters=5, lr=0.01, max_break_count=15,
                 break_threshold_ratio=0.001)
    model = Registration(cfg)
    model.load_pcds(np.zeros((20, 3), np.float32), np.zeros((20, 3), np.float32))
    with pytest.raises(_native.NdpError):
        model.register()
    for served in ("NSFP", "Nerfies", "ED"):                         # served baselines: same rule, GPU only
        base = Registration(Config(cfg, deformation_model=served))
        base.load_pcds(np.zeros((20, 3), np.float32), np.zeros((20, 3), np.float32))
        with pytest.raises(_native.NdpError):
            base.register()
    for other in ("Sinkhorn",):                                      # unserved comparison baseline (registration.py:123)
        with pytest.raises(KeyError):
            Registration(Config(cfg, deformation_model=other)).register()


def test_register_rejects_malformed_clouds_before_touching_the_device():
    """Empty / wrongly shaped clouds and mismatched landmark sets are reported as such (upstream dies inside knn_points)."""
    from deformationpyramid_amd.config import Config
    from deformationpyramid_amd.registration import Registration
    cfg = Config(deformation_model="NDP", device=torch.device("cpu"), depth=3, width=128, k0=-8, m=2, w_reg=0.0,
                 rotation_format="axis_angle", motion_type="SE3", samples=10, iters=5, lr=0.01, max_break_count=15,
                 break_threshold_ratio=0.001, w_ldmk=1.0, w_cd=0.0)
    ok = np.zeros((20, 3), np.float32)
    for src, tgt in ((np.zeros((0, 3), np.float32), ok), (ok, np.zeros((0, 3), np.float32)), (np.zeros((20, 2), np.float32), ok),
                     (np.zeros((20,), np.float32), ok)):
        model = Registration(cfg)
        model.load_pcds(src, tgt)
        with pytest.raises(ValueError, match="cloud must be"):
            model.register()
    model = Registration(cfg)
    model.load_pcds(ok, ok, landmarks=(torch.zeros(5, 3), torch.zeros(4, 3)))
    with pytest.raises(ValueError, match="landmarks"):
        model.register()


def test_config_files_and_join_constructor():
    from deformationpyramid_amd.config import load_config
    c = load_config(os.path.join(ROOT, "config", "NDP.yaml"), device=torch.device("cpu"))
    assert (c.m, c.k0, c.depth, c.width, c.samples, c.iters) == (9, -8, 3, 128, 2000, 500)
    assert c.motion_type == "SE3" and c.rotation_format == "axis_angle" and c.w_reg == 0.0
    assert c.split.test == "4DMatch-F" and c.snapshot_dir == "snapshot/pyramid_level/vis"
    l = load_config(os.path.join(ROOT, "config", "LNDP.yaml"), device=torch.device("cpu"))
    assert l.m == 10 and l.w_cd == 0.0 and l.trunc_cd == 0.25 and l.exp_dir == "0.3"
    # every key the hot path reads is present
    for k in ("deformation_model", "max_break_count", "break_threshold_ratio", "depth", "width", "k0", "m", "w_reg",
              "rotation_format", "motion_type", "samples", "lr", "iters"):
        assert k in c and k in l


def test_F8_flow_metrics_golden(golden):
    from deformationpyramid_amd.loss import compute_flow_metrics
    g = golden("F8_metrics")
    m = compute_flow_metrics(torch.from_numpy(g["flow"]), torch.from_numpy(g["flow_gt"]), torch.from_numpy(g["overlap"]))
    assert list(m.keys()) == list(g["keys"])
    np.testing.assert_allclose(np.array(list(m.values())), g["vals"], rtol=1e-4, atol=1e-4)


def test_metrics_nan_on_empty_subset_like_upstream():
    from deformationpyramid_amd.loss import compute_flow_metrics
    f = torch.rand(10, 3)
    m = compute_flow_metrics(f, f + 0.01, torch.ones(10, dtype=torch.bool))
    assert np.isnan(m["occ-epe"]) and not np.isnan(m["vis-epe"])


def test_timers_and_meters_protocol():
    from deformationpyramid_amd.utils import Timers, AverageMeter
    t = Timers()
    t.tic("registration"); t.toc("registration"); t.tictoc("backprop", 0.5)
    assert t.get_avg("backprop") == 0.5 and len(t.get_strings()) == 2
    a = AverageMeter()
    a.update(2.0); a.update(4.0)
    assert a.avg == 3.0 and a.count == 2


def test_synthetic_pair_matches_the_golden_generator(golden):
    from deformationpyramid_amd.synthetic import synthetic_pair
    g = golden("F7_end_to_end")
    src, tgt, flow_gt, overlap = synthetic_pair(5, n_total=2048)
    np.testing.assert_array_equal(src.numpy(), g["src"])
    np.testing.assert_array_equal(tgt.numpy(), g["tgt"])
    np.testing.assert_array_equal(flow_gt.numpy(), g["flow_gt"])
    np.testing.assert_array_equal(overlap.numpy(), g["overlap"])
    assert 0 < overlap.sum() < overlap.numel()


def test_surface_pair_matches_the_golden_generator(golden):
    """The product's surface-pair generator and the one the F10b fixture was captured with are the same function."""
    from deformationpyramid_amd.synthetic import surface_pair
    g = golden("F10b_surface_benchmark")
    src, tgt, flow_gt, overlap = surface_pair(0)
    np.testing.assert_array_equal(src[:16].numpy(), g["gen_src_head"])
    np.testing.assert_array_equal(tgt[:16].numpy(), g["gen_tgt_head"])
    np.testing.assert_array_equal(flow_gt[:16].numpy(), g["gen_flow_head"])
    assert [src.shape[0], tgt.shape[0], int(overlap.sum())] == list(g["gen_counts"])
    # the fixture is only worth something if the reference solves the problem: far better than doing nothing
    k = list(g["keys"])
    assert g["rows"][:, k.index("full-epe")].mean() < 0.6 * g["zero_flow_rows"][:, k.index("full-epe")].mean()
    assert g["rows"][:, k.index("full-AccS")].mean() > 25.0 > g["centroid_rows"][:, k.index("full-AccS")].mean()


def test_chamfer_argument_validation():
    from deformationpyramid_amd.loss import compute_truncated_chamfer_distance as cd
    with pytest.raises(ValueError):
        cd(torch.zeros(5, 3), torch.zeros(1, 5, 3))
    with pytest.raises(ValueError):
        cd(torch.zeros(1, 5, 3), torch.zeros(2, 5, 3))
    with pytest.raises(ValueError):
        cd(torch.zeros(1, 5, 3), torch.zeros(1, 5, 3), batch_reduction="max")
    with pytest.raises(ValueError):
        cd(torch.zeros(1, 5, 3), torch.zeros(1, 5, 3), weights=torch.tensor([-1.0]))


def test_two_rank_gloo_aggregation():
    """bench.py's pair sharding + SUM/MAX aggregation, world_size 2 on CPU (gloo)."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29517", os.path.join(ROOT, "tests", "_gloo_worker.py")],
                         capture_output=True, text=True, env=env, timeout=240)
    assert "AGG_OK" in out.stdout, out.stdout + out.stderr


def test_eight_rank_gloo_bench_aggregation():
    """The aggregation path bench.py runs at --gpus 8, on CPU: 8 gloo ranks, different pair counts, skewed elapsed times;
    whole-job value = sum of pairs / slowest rank, per-rank figures out of the SAME single all-reduce."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8",
                          "--master-addr", "127.0.0.1", "--master-port", "29523", os.path.join(ROOT, "tests", "_gloo_worker8.py")],
                         capture_output=True, text=True, env=env, timeout=600)
    assert "AGG8_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


def test_rank_affinity_plan_keeps_ranks_on_their_gpu_numa_node_and_apart():
    from deformationpyramid_amd.parallel import _parse_cpulist, plan_affinity
    assert _parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    topo = {0: list(range(0, 64)) + list(range(128, 192)), 1: list(range(64, 128)) + list(range(192, 256))}
    nodes = [0, 0, 0, 0, 1, 1, 1, 1]                       # 8 GPUs, 4 per socket
    sets = [plan_affinity(r, 8, nodes, lambda n: topo.get(n, []), set(range(256))) for r in range(8)]
    for r, cs in enumerate(sets):
        assert len(cs) == 32 and cs <= set(topo[nodes[r]])
    assert all(sets[a].isdisjoint(sets[b]) for a in range(8) for b in range(a))
    # restricted cgroup: only what is allowed is used; unknown topology: disjoint slices of the allowed set
    cs = plan_affinity(5, 8, nodes, lambda n: topo.get(n, []), set(range(60, 70)))
    assert cs and cs <= set(range(64, 70))
    sets = [plan_affinity(r, 4, [-1] * 4, lambda n: [], set(range(16))) for r in range(4)]
    assert [len(c) for c in sets] == [4] * 4 and set().union(*sets) == set(range(16))
    assert plan_affinity(0, 1, [3], lambda n: [], {7}) == {7}    # node known but no cpulist: stay where we are


def test_native_rng_replay_matches_torch():
    """libndp_host.so reproduces torch's CPU generator bit for bit (values and the state torch continues
    from), for the init replay of a whole pyramid and for mixed draw/discard sequences."""
    from deformationpyramid_amd import _native
    from deformationpyramid_amd.layout import LayerDesc
    from deformationpyramid_amd.nets import init_pyramid_store, _native_rng_ok
    assert _native_rng_ok()
    for descs in ([LayerDesc()] * 9, [LayerDesc(motion="Sim3", rotfmt="euler")] * 3,
                  [LayerDesc(rotfmt="quaternion"), LayerDesc(rotfmt="quaternion", nonrigidity=True)]):
        stride = (max(d.param_count for d in descs) + 63) // 64 * 64
        torch.manual_seed(5)
        a = init_pyramid_store(descs, 3, stride, native=False)
        ra = torch.randperm(1000)
        torch.manual_seed(5)
        b = init_pyramid_store(descs, 3, stride, native=True)
        rb = torch.randperm(1000)
        assert torch.equal(a, b) and torch.equal(ra, rb)
    # generator positions around the 624-word block boundary
    for skip in (0, 1, 622, 623, 624, 625, 1247):
        torch.manual_seed(11)
        torch.empty(skip).uniform_() if skip else None
        x = torch.empty(700).uniform_(-0.25, 1.75)
        torch.manual_seed(11)
        out = torch.zeros(700)
        _native.rng_replay([(skip, 0.0, 1.0, -1), (700, -0.25, 1.75, 0)] if skip else [(700, -0.25, 1.75, 0)], out)
        assert torch.equal(out, x), skip


def test_meshio_roundtrip_and_area_weighted_sampling(tmp_path):
    from deformationpyramid_amd.meshio import read_ply_ascii, sample_surface, write_ply_ascii
    # a unit square made of one big and one tiny triangle + a quad face (fan-triangulated on read)
    verts = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0], [2, 0, 0], [2, 0.01, 0]], dtype=np.float32)
    faces = np.array([[0, 1, 2], [1, 4, 5]], dtype=np.int64)
    path = tmp_path / "m.ply"
    write_ply_ascii(str(path), verts, faces)
    v, f = read_ply_ascii(str(path))
    np.testing.assert_array_equal(v, verts)
    np.testing.assert_array_equal(f, faces)
    with open(path, "a") as fh:                                  # append a quad: needs the header count bumped
        pass
    text = open(path).read().replace("element face 2", "element face 3") + "4 0 1 2 3\n"
    open(path, "w").write(text)
    v, f = read_ply_ascii(str(path))
    assert f.shape == (4, 3) and f[2].tolist() == [0, 1, 2] and f[3].tolist() == [0, 2, 3]
    pts = sample_surface(verts, faces, 4000, np.random.default_rng(0))
    assert pts.shape == (4000, 3) and pts.dtype == np.float32
    big = (pts[:, 0] <= 1.0).mean()                              # area 0.5 vs 0.005: ~99 % of the samples on the big one
    assert 0.97 < big <= 1.0
    assert np.all(pts[:, 2] == 0) and pts[:, 1].min() >= 0


def test_4dmatch_npz_reader_builds_ground_truth_like_upstream(tmp_path):
    """eval_nolearned.FourDMatchPairs: npz schema (_4dmatch.py:60-73) and GT flow / overlap (eval_nolearned.py:75-84)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("eval_nolearned", os.path.join(ROOT, "eval_nolearned.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    rng = np.random.default_rng(1)
    n, m = 50, 40
    s_pc, t_pc = rng.random((n, 3)).astype(np.float32), rng.random((m, 3)).astype(np.float32)
    flow = (0.05 * rng.standard_normal((n, 3))).astype(np.float32)
    ang = 0.3
    rot = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]], dtype=np.float32)
    trn = np.array([[0.1], [0.0], [-0.05]], dtype=np.float32)
    corr = np.stack([np.arange(0, 30), np.arange(0, 30)], 1)
    d = tmp_path / "4DMatch-F" / "seq"
    d.mkdir(parents=True)
    np.savez(d / "cam1_0000_cam2_0001.npz", s_pc=s_pc, t_pc=t_pc, s2t_flow=flow, rot=rot, trans=trn, correspondences=corr)
    ds = mod.FourDMatchPairs(str(tmp_path), "4DMatch-F")
    assert len(ds) == 1
    src, tgt, flow_gt, overlap = ds[0]
    want = (rot @ (s_pc + flow).T + trn).T - s_pc
    np.testing.assert_allclose(flow_gt.numpy(), want, rtol=0, atol=1e-6)
    assert overlap.sum().item() == 30 and overlap[:30].all() and not overlap[30:].any()
    assert src.shape == (n, 3) and tgt.shape == (m, 3)


def test_package_import_reserves_hardware_queues_for_its_streams():
    """Two engine streams on one ROCm hardware queue serialise: the package asks for 8 queues
    unless the user set the variable."""
    code = "import os; os.environ.pop('GPU_MAX_HW_QUEUES', None); import deformationpyramid_amd; print(os.environ['GPU_MAX_HW_QUEUES'])"
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT, timeout=300)
    assert out.returncode == 0 and out.stdout.strip() == "8", out.stderr[-500:]
    code = "import os; os.environ['GPU_MAX_HW_QUEUES'] = '3'; import deformationpyramid_amd; print(os.environ['GPU_MAX_HW_QUEUES'])"
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT, timeout=300)
    assert out.stdout.strip() == "3"


def test_bench_self_launch_builds_a_one_rank_per_gpu_command(monkeypatch):
    """`python bench.py --gpus 4` with no launcher re-executes itself under torch.distributed.run."""
    import subprocess
    import bench
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 0

    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "2"])
    with pytest.raises(SystemExit) as ex:
        bench.self_launch(4)
    assert ex.value.code == 0
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-4:] == ["--gpus", "4", "--steps", "2"] and cmd[-5].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_register_batch_rejects_mixed_landmark_batches():
    """Pairs with and without landmarks optimise different objectives (registration.py:189-212): one engine cannot
    serve both, and silently running them with the first pair's weights would be wrong."""
    from deformationpyramid_amd.config import load_config
    from deformationpyramid_amd.registration import Registration
    cfg = load_config(os.path.join(ROOT, "config", "LNDP.yaml"), device="cpu")
    a, b = torch.zeros(10, 3), torch.zeros(12, 3)
    with pytest.raises(ValueError, match="landmarks"):
        Registration(cfg).register_batch([(a, b), (a, b, (a[:4], b[:4]))])


def test_stale_library_is_detected_by_its_build_id(tmp_path):
    """The library carries the digest of its sources; the loader finds it without loading the file."""
    from deformationpyramid_amd import _native as N
    assert N._built_id(N.LIBPATH) == N.source_id()
    fake = tmp_path / "libfake.so"
    fake.write_bytes(b"\x7fELF....NDP_BUILD_ID=0123456789abcdef\x00....")
    assert N._built_id(str(fake)) == "0123456789abcdef" != N.source_id()
    L = N.lib()
    sizes = (ctypes.c_int * 6)()
    assert L.ndp_abi_sizes(sizes) == 0 and sizes[3] == ctypes.sizeof(N.Engine) and sizes[2] == ctypes.sizeof(N.PairState)


def test_two_way_fp16_split_model_is_as_exact_as_fp32():
    """The arithmetic of the split level kernels (csrc/ndp_fwd_split.inc), modelled in numpy: x = hi + 2^-11 lo with hi = fp16(x),
    lo = fp16(2^11 (x - hi)) represents x to within 2^-23 |x| -- one fp32 ulp -- and a 128-term contraction from the
    three products hi.hi + 2^-11 (hi.lo + lo.hi), accumulated in fp32, is at least as close to float64 as the fp32 chain.  An
    UNSCALED lo (fp16(x - hi)) is not: the remainder of a weight of 0.09 is a subnormal."""
    rng = np.random.default_rng(5)

    def split(x, scale=2048.0):
        hi = x.astype(np.float16).astype(np.float32)
        lo = ((x - hi).astype(np.float32) * np.float32(scale)).astype(np.float16).astype(np.float32)
        return hi, lo

    def mm32(a, b):                                   # exact products, fp32 accumulation in k-steps of 32 (the MFMA's)
        acc = np.zeros((a.shape[0], b.shape[1]), np.float32)
        for k0 in range(0, a.shape[1], 32):
            acc = (acc + (a[:, k0:k0 + 32].astype(np.float64) @ b[k0:k0 + 32].astype(np.float64)).astype(np.float32)).astype(np.float32)
        return acc

    W = rng.uniform(-0.088, 0.088, (128, 128)).astype(np.float32)
    H = np.maximum(rng.normal(0, 1, (128, 512)), 0).astype(np.float32)
    for x in (W, H):
        hi, lo = split(x)
        rep = hi.astype(np.float64) + lo.astype(np.float64) / 2048.0
        big = np.abs(x) >= 2.0 ** -11
        assert np.max(np.abs(rep - x)[big] / np.abs(x[big])) <= 2.0 ** -23        # one fp32 ulp at worst (fp32's own rounding: half of that)
        if (~big).any():
            assert np.max(np.abs(rep - x)[~big]) <= 2.0 ** -34                    # absolute floor for small values (subnormal steps of lo / 2^11)
    ref = W.astype(np.float64) @ H.astype(np.float64)
    chain = np.zeros_like(ref, dtype=np.float32)
    for k in range(128):
        chain = (chain + (W[:, k:k + 1] * H[k:k + 1, :]).astype(np.float32)).astype(np.float32)

    def three_products(scale):
        wh, wl = split(W, scale)
        hh, hl = split(H, scale)
        return (mm32(wh, hh) + (mm32(wh, hl) + mm32(wl, hh)).astype(np.float32) / np.float32(scale)).astype(np.float32)

    def rms(y):
        return float(np.sqrt(np.mean((y - ref) ** 2)))

    assert rms(three_products(2048.0)) <= rms(chain)
    assert rms(three_products(1.0)) > 1.5 * rms(chain)


@pytest.mark.parametrize("n_src,n_tgt,keep,seed", [(8192, 8192, 2000, 0), (1, 5, 2000, 3), (2001, 777, 2000, 11), (24856, 24856, 6000, 5), (300, 300, 0, 7)])
def test_native_pair_init_replays_torch_randperm_and_leaves_the_generator_where_torch_would(n_src, n_tgt, keep, seed):
    """ndp_pair_init (init draws + the prefixes of BOTH sampling permutations from a generator-state snapshot) and ndp_rng_skip (the
    generator state after a pair) against torch itself: the same uniform_ stream, the same torch.randperm(n)[:keep] prefixes, and a
    generator that continues exactly where torch's would -- for full-size, tiny, ragged and shape-transfer sizes and keep = 0."""
    import ctypes
    from deformationpyramid_amd import _native as N
    from deformationpyramid_amd.nets import _draw_ops, Deformation_Pyramid
    L = N.host_lib()
    torch.manual_seed(seed)
    pyr = Deformation_Pyramid(depth=3, width=128, device="cpu", k0=-8, m=3, rotation_format="axis_angle", motion="SE3")   # torch path: init draws
    want_s = torch.randperm(n_src)[:keep]
    want_t = torch.randperm(n_tgt)[:keep]
    after = torch.get_rng_state().clone()
    follow = torch.rand(4)
    # native replay from the same seed
    torch.manual_seed(seed)
    st = torch.get_rng_state().clone()
    descs = pyr.descs
    stride = pyr.store.shape[1]
    ops_ = N.make_draw_ops(_draw_ops(descs, 3, stride))
    n_par = 3 * stride
    host = torch.zeros(n_par + 2 * max(keep, 1) + 8)
    hi = host[n_par:].view(torch.int32)
    scratch = torch.zeros(max(n_src, n_tgt) + 8, dtype=torch.int64)
    rc = L.ndp_pair_init(ctypes.c_void_p(st.data_ptr()), st.numel(), ops_, len(ops_), ctypes.c_void_p(host.data_ptr()), n_src, n_tgt, keep,
                         ctypes.c_void_p(hi.data_ptr()), ctypes.c_void_p(hi[keep:].data_ptr()), ctypes.c_void_p(scratch.data_ptr()))
    assert rc == 0
    store = host[:n_par].view(3, stride)
    for i, d in enumerate(descs):
        assert torch.equal(store[i, :d.param_count], pyr.store[i, :d.param_count]), i
    ks, kt = min(keep, n_src), min(keep, n_tgt)
    assert torch.equal(hi[:ks].long(), want_s[:ks]) and torch.equal(hi[keep:keep + kt].long(), want_t[:kt])
    assert L.ndp_rng_skip(ctypes.c_void_p(st.data_ptr()), st.numel(), L.ndp_pair_draws(ops_, len(ops_), n_src, n_tgt)) == 0
    assert torch.equal(st, after)
    torch.set_rng_state(st)
    assert torch.equal(torch.rand(4), follow)
