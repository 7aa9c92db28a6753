"""GPU test of the data-parallel path on a ONE-GPU box: two processes share cuda:0 and exchange the
flat gradient bucket over gloo (the production backend is nccl = RCCL; the code path -- bucket slots written in
place by the HIP layers, one all-reduce, grad_scale folded into the fused optimizer -- is the same).
Claim checked: 2 ranks x half batch, SUM all-reduce, scale 1/world  ==  1 process x full batch."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build(hip, seed=123):
    import neunet_hip.nn as nn
    np.random.seed(seed)

    class MLP(nn.Module):
        def __init__(self):
            super().__init__()
            self.l1 = nn.Linear(64, 96)
            self.act = nn.Swish()
            self.norm = nn.RMSNorm(96)
            self.l2 = nn.Linear(96, 10)

        def forward(self, x):
            return self.l2(self.norm(self.act(self.l1(x))))

    return MLP()


def _data():
    rng = np.random.default_rng(9)
    return rng.uniform(-1, 1, (3, 32, 64)).astype(np.float32), rng.integers(0, 10, (3, 32)).astype(np.int32)


def _run(rank, world, port, q, overlap=False):
    import neunet_hip as hip
    import neunet_hip.nn as nn
    from neunet_hip.distributed import GradBucket, shard_batch
    from neunet_hip.optim import AdamW
    if world > 1:
        import torch.distributed as dist
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    model = _build(hip)
    params = model.parameters()
    bucket = GradBucket(params, overlap=overlap, segment_bytes=1 << 12)   # overlap: several segments, async all-reduce
    opt = AdamW(params, lr=1e-2, weight_decay=1e-2)
    opt.grad_scale = 1.0 / world
    loss_fn = nn.CrossEntropyLoss()
    X, Y = _data()
    lo, hi = shard_batch(32, rank, world)
    for s in range(3):
        opt.zero_grad()
        out = model(hip.Tensor(X[s, lo:hi], device="cuda", requires_grad=overlap))   # overlap: dW before dX in every Linear
        loss_fn(out, hip.Tensor(Y[s, lo:hi], dtype=np.int32, requires_grad=False, device="cuda")).backward()
        bucket.all_reduce()
        opt.step()
    res = [p.numpy().copy() for p in params]
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()
    if q is not None:
        q.put((rank, res))
    return res


@pytest.mark.parametrize("overlap", [False, True])
def test_dp2_equals_full_batch(overlap):
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible")
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_run, args=(r, 2, port, q, overlap)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    ref = _run(0, 1, 0, None)
    for a, b, r in zip(got[0], got[1], ref):
        np.testing.assert_array_equal(a, b)                         # replicas stay bit-identical
        np.testing.assert_allclose(a, r, rtol=1e-4, atol=1e-5)      # and equal the single-process full batch


# ---- hipGraph-replayed DP step: 'sum' loss per rank, device-side target count in the bucket, divisor inside Adam ----------
def _run_graphed(rank, world, port, q, overlap, backend="gloo", graphed=True, force=False, ingraph=False, recipe=None):
    """The bench's C4 data-parallel recipe at toy size.  world > 1: every rank back-propagates CrossEntropy(reduction='sum',
    ignore_index=0) on its shard, a kernel writes the shard's non-ignored count into the bucket's extra slot, the bucket
    (cut into segments, exchanged asynchronously between the pieces of the captured backward pass when overlap=True) is
    SUM-all-reduced, and the fused optimizer divides by the all-reduced count -- no host read anywhere.
    world == 1: plain 'mean' loss, eager or graphed.  Sequence: 1 warm-up step on batch 0, then batches 0, 1, 2."""
    import neunet_hip as hip
    import neunet_hip.nn as nn
    from neunet_hip._lib import call_hip_function, get_current_stream_ptr
    from neunet_hip.distributed import GradBucket, shard_batch
    from neunet_hip.graph import GraphedTrainStep
    from neunet_hip.optim import AdamW
    from neunet_hip import distributed as D
    dev = rank if (backend == "nccl" and world > 1) else 0
    torch.cuda.set_device(dev)
    info = {}
    if world > 1 or force:
        import torch.distributed as dist
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        os.environ["RANK"], os.environ["WORLD_SIZE"] = str(rank), str(world)
        D.init_process_group(backend, force=force)      # force: a ONE-rank group whose collectives are issued all the same
        info["backend"] = dist.get_backend()
        info["live"] = D.collectives_live()
    # recipe "dp": 'sum' loss + device-side count + divisor inside Adam (what every DP rank runs); "plain": 'mean' loss
    dp = (world > 1 or force) if recipe is None else recipe == "dp"
    model = _build(hip)
    params = model.parameters()
    bucket = GradBucket(params, extra_scalars=1, overlap=overlap and (world > 1 or force), segment_bytes=1 << 12)
    opt = AdamW(params, lr=1e-2, weight_decay=1e-2)
    loss_fn = nn.CrossEntropyLoss(ignore_index=0, reduction="sum" if dp else "mean")
    if dp:
        opt.grad_divisor = bucket.extra
    X, Y = _data()
    lo, hi = shard_batch(32, rank, world)
    xs = hip.Tensor(X[0, lo:hi], device="cuda", requires_grad=True)
    ys = hip.Tensor(Y[0, lo:hi], dtype=np.int32, requires_grad=False, device="cuda")

    def fb():
        xs.grad = None
        loss = loss_fn(model(xs), ys)
        if dp:
            call_hip_function("nnhipCrossEntropyDenominator", ys.data, 4, ys.data.numel(), 0, None, 10, None, bucket.extra,
                              get_current_stream_ptr())
        loss.backward()
        return loss

    def eager():
        opt.zero_grad()
        fb()
        bucket.all_reduce()
        opt.step()

    if graphed:
        step = GraphedTrainStep(fb, opt, bucket, warmup=1, world=world, capture_collectives=ingraph)
        n_pieces = len(step.pieces)
        info["mode"], info["ingraph_error"] = step.mode, step.ingraph_error
    else:
        eager()
        step, n_pieces = eager, 0
    for s in range(3):
        xs.data.copy_(torch.from_numpy(X[s, lo:hi]))
        ys.data.copy_(torch.from_numpy(Y[s, lo:hi]))
        if s == 1 and graphed:
            opt.lr = 5e-3            # an LR change between replays must take effect (device-side hyper-parameters)
        elif s == 1:
            opt.lr = 5e-3
        step()
    torch.cuda.synchronize()
    res = [p.numpy().copy() for p in params]
    if graphed:
        step.release()
    if world > 1 or force:
        import torch.distributed as dist
        dist.destroy_process_group()
        D.force_collectives = False
    if q is not None:
        q.put((rank, res, n_pieces, info))
    return res


def _spawn2(target, args):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, 2, port, q) + args) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return {g[0]: g[1:] for g in got}


@pytest.mark.parametrize("overlap", [False, True])
def test_dp2_graphed_sum_loss_with_device_count_equals_full_batch_mean(overlap):
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible")
    got = _spawn2(_run_graphed, (overlap,))
    ref = _run_graphed(0, 1, 0, None, False, graphed=False)          # one process, full batch, eager, 'mean' loss
    ref_g = _run_graphed(0, 1, 0, None, False, graphed=True)         # ... and the same through one captured graph
    if overlap:
        assert got[0][1] > 1                                         # the backward pass really was cut into pieces
    for a, b, r, rg in zip(got[0][0], got[1][0], ref, ref_g):
        np.testing.assert_array_equal(a, b)                          # replicas stay bit-identical
        np.testing.assert_allclose(a, r, rtol=1e-4, atol=1e-5)       # and equal the single-process full batch
        np.testing.assert_allclose(rg, r, rtol=1e-5, atol=1e-6)      # graph replay (incl. the LR change) == eager


def test_dp2_nccl_when_two_gpus_are_visible():
    """The same graphed, overlapped step over RCCL (backend nccl), one GPU per rank.  Skips on a 1-GPU box."""
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible")
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs (RCCL over xGMI); the driver's multi-GPU box runs it")
    got = _spawn2(_run_graphed, (True, "nccl"))
    ref = _run_graphed(0, 1, 0, None, False, graphed=False)
    for a, b, r in zip(got[0][0], got[1][0], ref):
        np.testing.assert_array_equal(a, b)
        np.testing.assert_allclose(a, r, rtol=1e-4, atol=1e-5)


# ---- the RCCL leg on a ONE-GPU box: a 1-rank `nccl` process group with the collectives forced on ---------------------------
def _spawn1(target, args):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=target, args=(0, 1, _free_port(), q) + args)
    p.start()
    got = q.get(timeout=300)
    p.join(timeout=120)
    assert p.exitcode == 0
    return got[1:]


@pytest.mark.parametrize("launch", ["eager", "pieces", "ingraph"])
def test_dp1_forced_nccl_runs_the_dp_machinery_on_rccl(launch):
    """Every DP test above exchanges over gloo; the production backend is nccl (= RCCL).  Here ONE rank makes an `nccl`
    process group and `force_collectives` makes the bucket issue its per-segment asynchronous all-reduces on RCCL's
    stream anyway: eager, between the pieces of the captured backward pass, and captured INTO the step graph.
    Claims: the backend really is nccl; the overlapped step is cut into > 1 pieces; parameters after 3 steps are
    BIT-IDENTICAL to the same 'sum'-loss + device-divisor recipe run with no process group at all (a 1-rank SUM
    all-reduce is the identity, so any difference is a stream-ordering bug), and within 1e-4 of the plain 'mean' step."""
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible")
    res, n_pieces, info = _spawn1(_run_graphed, (True, "nccl", launch != "eager", True, launch == "ingraph"))
    assert info["backend"] == "nccl" and info["live"] is True
    if launch == "pieces":
        assert info["mode"] == "pieces" and n_pieces > 1
    if launch == "ingraph":
        assert info["mode"] in ("ingraph", "pieces"), info      # a refused capture must have fallen back, and said why
        assert info["mode"] == "ingraph" or info["ingraph_error"]
    same = _run_graphed(0, 1, 0, None, False, graphed=False, recipe="dp")       # same recipe, no process group
    plain = _run_graphed(0, 1, 0, None, False, graphed=False)                   # 'mean' loss, the non-DP step
    for a, b, c in zip(res, same, plain):
        np.testing.assert_array_equal(a, b)
        np.testing.assert_allclose(a, c, rtol=1e-4, atol=1e-5)


def _run_gpt_dp(rank, world, port, q, launch, force):
    """GPT-tiny (the C4 graph: embedding, fused q|k|v attention blocks with grouped bucket slots, RMSNorm, FFN, vocab
    head, CE with PAD ignored) at d128 L2 through bench.py's DP recipe.  launch: eager / pieces / ingraph."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
    import gpt_tiny
    import neunet_hip as hip
    import neunet_hip.nn as nn
    from neunet_hip import distributed as D
    from neunet_hip._lib import call_hip_function, get_current_stream_ptr
    from neunet_hip.graph import GraphedTrainStep
    from neunet_hip.optim import Adam
    torch.cuda.set_device(0)
    info = {}
    comm = None
    if force == "native":                                  # RCCL through the library's own C ABI: no torch process group
        os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
        comm = D.NativeComm.from_env()
        D.force_collectives = True
        info["backend"] = "native:" + D.NativeComm.library()[0]
        info["rccl_version"] = D.NativeComm.library()[1]
    elif force:
        import torch.distributed as dist
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
        D.init_process_group("nccl", force=True)
        info["backend"] = dist.get_backend()
    V, Dm, H, F, L, B, T = 1000, 128, 4, 512, 2, 8, 64
    rng = np.random.default_rng(77)
    batches = []
    for _ in range(3):
        b = rng.integers(3, V, (B, T + 1)).astype(np.int32)
        b[1, -9:] = 0
        batches.append(b)
    np.random.seed(1004)
    model = gpt_tiny.build_gpt(V, Dm, H, F, L, pad_idx=0, max_len=256, fused=True)
    ids = hip.Tensor(np.ascontiguousarray(batches[0][:, :-1]), dtype=np.int32, requires_grad=False, device="cuda")
    tgt = hip.Tensor(np.ascontiguousarray(batches[0][:, 1:]).reshape(-1), dtype=np.int32, requires_grad=False, device="cuda")
    loss_fn = nn.CrossEntropyLoss(ignore_index=0, reduction="sum")
    state = {"bucket": None}

    def fb():
        out, _ = model.forward(ids)
        loss = loss_fn(out.reshape(B * T, V), tgt)
        if state["bucket"] is not None:
            call_hip_function("nnhipCrossEntropyDenominator", tgt.data, 4, tgt.data.numel(), 0, None, V, None,
                              state["bucket"].extra, get_current_stream_ptr())
        loss.backward()
        return loss

    fb()
    active = [p for p in model.parameters() if p.grad is not None]
    opt = Adam(model.parameters(), lr=1.5e-4, betas=(0.9, 0.98), eps=1e-9)
    opt.zero_grad()
    bucket = D.GradBucket(active, extra_scalars=1, overlap=bool(force), segment_bytes=1 << 18, comm=comm)
    state["bucket"] = bucket
    opt.grad_divisor = bucket.extra

    def feed(b):
        ids.data.copy_(torch.from_numpy(np.ascontiguousarray(b[:, :-1])))
        tgt.data.copy_(torch.from_numpy(np.ascontiguousarray(b[:, 1:]).reshape(-1)))

    def eager():
        opt.zero_grad()
        fb()
        bucket.all_reduce()
        opt.step()

    if launch == "eager":
        eager()                                            # the graphed runs warm up once on batch 0
        step = eager
    else:
        step = GraphedTrainStep(fb, opt, bucket, warmup=1, world=1, capture_collectives=launch == "ingraph")
        info.update(mode=step.mode, pieces=len(step.pieces), ingraph_error=step.ingraph_error,
                    segments=len(bucket.segments))
    for b in batches:
        feed(b)
        step()
    torch.cuda.synchronize()
    res = [p.numpy().copy() for p in model.parameters()]
    if launch != "eager":
        step.release()
    if comm is not None:
        info["live"] = D.collectives_live(comm)
        comm.destroy()
        D.force_collectives = False
    elif force:
        import torch.distributed as dist
        dist.destroy_process_group()
        D.force_collectives = False
    if q is not None:
        q.put((rank, res, info))
    return res


@pytest.mark.parametrize("launch", ["eager", "pieces", "ingraph"])
def test_gpt_forced_nccl_dp_step_is_bit_identical_to_the_local_step(launch):
    """The GPT step through GradBucket(overlap) + per-segment graph cuts + async all_reduce on RCCL's stream (1-rank nccl,
    collectives forced) leaves every parameter BIT-IDENTICAL to the same recipe with no process group: a 1-rank SUM
    all-reduce is the identity, so this pins the stream ordering between our kernels and the backend's stream, the
    graph cuts, and the device-side divisor -- on the production backend."""
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible")
    res, info = _spawn1(_run_gpt_dp, (launch, True))
    assert info["backend"] == "nccl"
    if launch == "pieces":
        assert info["mode"] == "pieces" and info["segments"] > 1 and info["pieces"] == info["segments"] + 1, info
    if launch == "ingraph":
        assert info["mode"] == "ingraph" or info["ingraph_error"], info
    local = _run_gpt_dp(0, 1, 0, None, "eager", False)
    for a, b in zip(res, local):
        np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("launch", ["eager", "pieces", "ingraph"])
def test_gpt_native_comm_dp_step_is_bit_identical_to_the_local_step(launch):
    """The same GPT DP step with the exchange going through the library's OWN RCCL entry points (nnhipCommInitRank /
    nnhipAllReduceSumF32 behind `NativeComm`, SURVEY 8b "add AllReduce*") instead of torch.distributed: a 1-rank communicator,
    collectives forced, per-segment all-reduces on the communicator's side stream -- eager, between graph pieces, and captured
    into the step graph.  Bit-identical to the recipe without any communicator."""
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible")
    res, info = _spawn1(_run_gpt_dp, (launch, "native"))
    assert info["backend"].startswith("native:") and "rccl" in info["backend"] and info["live"] is True, info
    if launch == "pieces":
        assert info["mode"] == "pieces" and info["segments"] > 1 and info["pieces"] == info["segments"] + 1, info
    if launch == "ingraph":
        assert info["mode"] == "ingraph" or info["ingraph_error"], info
    local = _run_gpt_dp(0, 1, 0, None, "eager", False)
    for a, b in zip(res, local):
        np.testing.assert_array_equal(a, b)


def _native_collectives(rank, world, port, q):
    from neunet_hip import distributed as D
    from neunet_hip._lib import NeunetHipError
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    comm = D.NativeComm.from_env()
    x = torch.arange(1, 1025, dtype=torch.float32, device="cuda") * (rank + 1)
    ref = x.clone()
    out = {}
    comm.all_reduce(x)                                    # SUM over ranks of (rank+1) * v
    out["sum"] = x.cpu().numpy().copy()
    y = ref.clone()
    w = comm.all_reduce(y, op="avg", async_op=True)       # side stream + join
    w.wait()
    out["avg"] = y.cpu().numpy().copy()
    z = ref.clone()
    comm.broadcast(z, root=0)
    out["bcast"] = z.cpu().numpy().copy()
    try:
        comm.all_reduce(torch.zeros(4, dtype=torch.float64, device="cuda"))
        out["typeerr"] = False
    except TypeError:
        out["typeerr"] = True
    comm.destroy()
    try:
        comm.all_reduce(x)
        out["dead"] = False
    except RuntimeError:
        out["dead"] = True
    q.put((rank, out, D.NativeComm.library()))


def test_native_comm_collectives_one_rank():
    """nnhipCommUniqueId -> nnhipCommInitRank -> nnhipAllReduceSumF32 / AvgF32 / BroadcastF32 -> nnhipCommDestroy on a 1-rank
    communicator: every collective is the identity, the handle dies loudly, the bound library is an RCCL."""
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible")
    out, lib = _spawn1(_native_collectives, ())
    v = np.arange(1, 1025, dtype=np.float32)
    for k in ("sum", "avg", "bcast"):
        np.testing.assert_array_equal(out[k], v)
    assert out["typeerr"] and out["dead"]
    assert "rccl" in lib[0] and lib[1] > 20000, lib


def test_native_comm_two_ranks_when_two_gpus_are_visible():
    """Two ranks, two GPUs, the id shipped through a TCPStore (no torch process group): SUM / AVG / broadcast values."""
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible")
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (the driver's multi-GPU tier)")
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q, port = ctx.Queue(), _free_port()
    ps = [ctx.Process(target=_native_collectives, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    got = dict((r, o) for r, o, _ in (q.get(timeout=300) for _ in ps))
    for p in ps:
        p.join(timeout=120)
        assert p.exitcode == 0
    v = np.arange(1, 1025, dtype=np.float32)
    for r in range(2):
        np.testing.assert_array_equal(got[r]["sum"], 3 * v)
        np.testing.assert_array_equal(got[r]["avg"], 1.5 * v)
        np.testing.assert_array_equal(got[r]["bcast"], v)


def test_bench_eight_ranks_on_one_gpu_rehearsal():
    """First-contact rehearsal of the run the driver makes on an 8-GPU node (round-4 review, item 5): `bench.py --gpus 8` at toy
    sizes (NNHIP_BENCH_TOY=1), eight ranks sharing the one visible GPU over gloo (NNHIP_ALLOW_OVERSUBSCRIBE=1).  Covered: the
    self-launch under torch.distributed.run with a free port on 127.0.0.1, the ONE JSON line on stdout with every key the driver
    reads, strong scaling with a global batch that does NOT divide over the ranks (12 sequences on 8 ranks: shard_batch), the
    other curve under also.c4_weak, the overlapped bucket segments, and the in-graph capture request falling back to graph
    pieces with the reason recorded (gloo collectives cannot be captured)."""
    import json
    import subprocess
    import sys
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, NNHIP_ALLOW_OVERSUBSCRIBE="1", NNHIP_BENCH_TOY="1", NNHIP_BENCH_C3="0", NNHIP_BENCH_C5="0",
               NNHIP_BENCH_NB="0", NNHIP_BENCH_BF16X3="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "4", "--warmup", "2", "--c4-batch", "12",
                        "--scaling", "strong", "--dp-ingraph", "1", "--c1-steps", "32", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, f"stdout must carry exactly one JSON line, got {len(lines)}: {r.stdout[:500]}"
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "rccl_ranks"):
        assert key in d, key
    assert d["n_gpus"] == 8 and d["rccl_ranks"] == 8 and d["steps"] == 4 and d["scaling"] == "strong"
    assert d["config"]["global_batch"] == 12 and d["config"]["parallelism"] == "dp8" and "TOY" in d["config"]["workload"]
    assert d["value"] > 0 and abs(d["value"] - 12 / (d["ms_per_step"] * 1e-3)) < 1e-2 * d["value"]
    assert d["dist_backend"] == "gloo"
    assert "segments" in d["dp_exchange"] or "all-reduce" in d["dp_exchange"]
    mode = d["dp_mode"]
    assert mode["overlap"] in (True, False) and mode["launch"] is not None
    assert mode["ingraph_error"] or mode["modes_that_failed"] or "graph" in str(mode["launch"]), mode      # the capture request was answered
    weak = d["also"]["c4_weak"]
    assert weak["global_batch"] == 96 and weak["samples_per_s"] > 0
    assert d["also"]["c1"].get("samples_per_s", 0) > 0 or "error" in d["also"]["c1"]
    assert "bound" in d["roofline"] and "frac" in d["roofline"]
    # round 6: the first multi-rank record has to explain itself -- the exchange timed on its own and what it adds to the step,
    # RCCL's own account of its algorithm / protocol choice (gloo has none: the key is there and says so), the GEMM mode
    ar = d["allreduce_ms"]
    assert "error" not in ar, ar
    assert ar["total"] > 0 and ar["exposed"] is not None and ar["exposed"] >= 0
    assert len(ar["pieces"]) >= 1 and all(p_["ms_alone"] > 0 for p_ in ar["pieces"])
    assert abs(ar["total"] - sum(p_["ms_alone"] for p_ in ar["pieces"])) < 1e-2
    assert ar["ms_per_step_collectives_muted"] > 0
    assert "rccl" in d and ("choices" in d["rccl"] or "note" in d["rccl"])
    assert d["gemm_mode"] == 0 and d["dtype"] == "f32"
