#!/usr/bin/env python3
"""Headline benchmark: encode+decode throughput (Mpix/s) of the hyperprior codec (BASELINE.json configs[4] shape:
synthetic 3x256x256 images, N=128 / M=192, configs/lossy_graph_scalable_exp_hp.py) on one MI355X per process.

  python bench.py --gpus N --steps K --warmup W
      N == 1: runs in this process.  N > 1 without a launcher: this process starts
      `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same flags>` as a CHILD before it has
      touched the GPU and exits with the child's code (one rank per GPU, RCCL over xGMI for the metric reduction only).

A "step" = every image of the rank's batch goes through codec.compress() and codec.decompress() once.  The K steps are
spread over `--workers` concurrent stream workers (cbench_basic_amd/benchmark/stream_workers.py, the analogue of the
reference's num_testing_workers pool, basic_benchmark.py:829-858, which hands whole dataset items to its workers): step
k runs on worker k mod W, each worker loops compress -> decompress over the whole batch on its own HIP stream, so W
batches are in flight and one worker's serial rANS chains run beside another worker's MFMA transforms
(`--shard-by images` cuts every step's batch into W shards instead).  The timed region is bracketed by barrier +
synchronize on both sides and holds exactly K steps; `config.call_latency_ms` is the wall time of one
compress+decompress call of one worker.

`value` is measured with the input batch resident in HBM (the tier's contract); `pcie_inclusive` repeats the run with
the batch in page-locked host memory, uploaded INSIDE compress() as the reference's timed region does
(general_codec.py:46-47 under basic_benchmark.py:199-202).  `scaling` is "weak" (--batch images per GPU at every N);
at N > 1 the same run also times BASELINE configs[4] as written (256 images over N GPUs) and reports it as `strong`.

One JSON line on rank 0, with
  roofline     -- MFMA transform kernels: algorithmic FLOPs / HIP-event time of the transform launches vs 157.3 TF fp32,
                  plus `end_to_end` = FLOPs of a step / wall time of a step (everything included)
  cpu_baseline -- the CPU oracle (PyTorch-CPU fp32 convs + C rANS restatement) on a bounded sample of the SAME images
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md "Peak FP32 (matrix)"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=256, help="images per GPU per step (weak scaling)")
    ap.add_argument("--total", type=int, default=256, help="images per step over ALL GPUs in the strong-scaling leg (cfg-5)")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--workers", type=int, default=None,
                    help="concurrent stream workers per GPU (1 = plain sequential calls); default 6 (headline and AR workloads)")
    ap.add_argument("--shard-by", default="steps", choices=["steps", "images"],
                    help="how the K steps are spread over the workers: steps = every worker codes WHOLE batches, step k on worker k mod W "
                         "(W batches in flight; the reference's pool also hands whole dataset items to its workers); "
                         "images = every step's batch is cut into W contiguous shards")
    ap.add_argument("--rans-waves", type=int, default=-1, help="image streams per rANS workgroup (-1: 8 for whole batches in flight, 4 for image shards, library default with one worker)")
    ap.add_argument("--token-lanes", type=int, default=None,
                    help="transform phases of the workers admitted at a time (1: full batches; 2: small batches, whose launches leave compute units idle); default by batch size")
    ap.add_argument("--strong-workers", type=int, default=6,
                    help="stream workers of the strong_per_gpu_proxy leg (total/8 images per step = one rank's share of the N = 8 strong-scaling leg)")
    ap.add_argument("--strong-in-process", action="store_true", help="N = 1: also run the strong leg's per-GPU share inside this process (the N > 1 code path)")
    ap.add_argument("--strong-first", action="store_true", help="run the strong leg before the weak legs (experiment)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="bound of the CPU-baseline sample of the AR workload lines")
    ap.add_argument("--no-ar-workloads", action="store_true", help="do not add the ar_workloads lines (child processes) to the N = 1 line")
    ap.add_argument("--cpu-images", type=int, default=2000, help="bounded CPU-baseline sample (images)")
    ap.add_argument("--input", default="hbm", choices=["hbm", "host"],
                    help="where the batch lives when the timed region starts: hbm = resident (the tier's contract for `value`), "
                         "host = page-locked host memory, uploaded inside compress() (profiling the pcie_inclusive leg on its own)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the pcie_inclusive and strong-scaling legs (profiling runs)")
    ap.add_argument("--no-dominant", action="store_true", help="skip the single-launch roofline (counter passes: keeps the launch mix = timed passes)")
    ap.add_argument("--levels", type=int, nargs="*", default=None, help="--workload basic: complexity levels to time (default 0 3 7; the first is the line's value)")
    ap.add_argument("--no-kodak-leg", action="store_true", help="--workload basic: skip the Kodak-shaped batch-1 leg through the harness (a child process)")
    ap.add_argument("--workload", default="hyperprior", choices=["hyperprior", "checkerboard", "basic"],
                    help="hyperprior = the headline (BASELINE configs[4] shape); checkerboard = configs[2] topo-group AR codec; "
                         "basic = configs[3] BaSIC slimmable scan-line codec at complexity level 0 (parity-test configurations, extra lines)")
    ap.add_argument("--master-port", type=int, default=29533)
    ap.add_argument("--dist-backend", default="nccl", help="torch.distributed backend of the metric reduction ('nccl' IS RCCL on ROCm)")
    ap.add_argument("--same-device", action="store_true",
                    help="REHEARSAL ONLY: every rank uses GPU 0 (with --dist-backend gloo: RCCL refuses two ranks on one GPU), to walk the "
                         "N > 1 code path on a one-GPU box; the line it prints is not a measurement")
    return ap.parse_args()


def self_launch(args):
    """--gpus N > 1 outside a launcher: become the parent of a torch.distributed.run job.  Nothing in this process has
    initialised the GPU (torch is not even imported yet), and the job runs as a CHILD -- never an exec."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(args.master_port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def image(i, size):
    """Image i of the synthetic set: torch.manual_seed(i); torch.rand(3, S, S)
    (configs/datasets/images/random_image_generator.py:12-15) -- the SAME images feed the GPU legs and the CPU baseline."""
    import torch
    torch.manual_seed(int(i))
    return torch.rand(3, size, size)


def conv_flops_per_image(codec, size):
    """Algorithmic FLOPs (2*MAC, conv + GDN) of encode (g_a, h_a, h_s) + decode (h_s, g_s) per image."""
    ec = codec.entropy_coder
    g_a, h_a = ec.latent_inference_modules["x_y"], ec.latent_inference_modules["y_z"]
    h_s, g_s = ec.latent_generative_modules["z_y"], ec.latent_generative_modules["y_x"]
    s16, s64 = size // 16, size // 64
    enc = g_a.flops(1, size, size) + h_a.flops(1, s16, s16) + h_s.flops(1, s64, s64)
    dec = h_s.flops(1, s64, s64) + g_s.flops(1, s16, s16)
    return enc, dec


def measure_conv_kernels(codec, x, reps=3):
    """HIP-event time of the transform launches alone (events on the stream the kernels are launched on), returning
    (seconds per pass over the batch, launches per pass)."""
    import torch
    ec = codec.entropy_coder
    g_a, h_a = ec.latent_inference_modules["x_y"], ec.latent_inference_modules["y_z"]
    h_s, g_s = ec.latent_generative_modules["z_y"], ec.latent_generative_modules["y_x"]

    def one_pass():
        y = g_a(x)
        z = h_a(y)
        h_s(z)
        h_s(z)
        return g_s(y)

    one_pass()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        one_pass()
    e1.record()
    torch.cuda.synchronize()

    def count(m, t):
        n, (b, _, h, w) = 0, t.shape
        for p in m.plans():
            n += p.launches(b, h, w)
            h, w = p.out_hw(h, w)
        return n
    y = g_a(x)
    z = h_a(y)
    launches = count(g_a, x) + count(h_a, y) + 2 * count(h_s, z) + count(g_s, y)
    return e0.elapsed_time(e1) / 1e3 / reps, launches


def measure_dominant_kernel(codec, x, reps=5):
    """The single heaviest launch of the pass -- the second analysis layer (conv 5x5 s2 128->128 + GDN on the
    H/2 x W/2 map), one conv_tap_mfma_kernel<4,4,5,5,8> launch -- timed alone with HIP events."""
    import torch
    g_a = codec.entropy_coder.latent_inference_modules["x_y"]
    p0, p1 = g_a.plans()[0], g_a.plans()[1]
    h1 = p0(x)
    y = p1(h1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        p1(h1, out=y)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    fl = p1.flops(h1.shape[0], h1.shape[2], h1.shape[3])
    return dict(kernel="conv_tap_mfma_kernel<4,4,5,5,8> (g_a layer 2: conv5x5 s2 128->128 + GDN, one launch)",
                flops_per_launch=fl, avg_launch_ms=ms, achieved=fl / ms / 1e9, frac=fl / ms / 1e9 / PEAK_FP32_MFMA_TFLOPS)


def cpu_baseline(codec_cpu_state, n_images, size):
    import torch
    from oracle.codec_oracle import HyperpriorOracle
    # the GPU box gives one GPU a 16-core CPU share; more threads than that only oversubscribes
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))
    torch.set_num_threads(cores)
    oracle = HyperpriorOracle(codec_cpu_state)
    oracle.decompress(oracle.compress(image(0, size).unsqueeze(0)))  # warm-up
    t0, done = time.time(), 0
    for i in range(n_images):  # bounded sample: stop after ~12 s of CPU work
        x = image(i, size).unsqueeze(0)
        oracle.decompress(oracle.compress(x))
        done += 1
        if time.time() - t0 > 12.0:
            break
    dt = time.time() - t0
    return dict(value=done * size * size / dt / 1e6, unit="Mpix/s", cores=cores, kind="port",
                sample=f"images 0..{done - 1} of the same synthetic set (3x{size}x{size}), batch 1, PyTorch-CPU fp32 transforms + "
                       f"C rANS oracle, {dt:.1f} s wall")


def traffic_from_profiles():
    """HBM bytes per transform launch from the SEPARATE rocprofv3 --pmc passes committed under profiles/ (a profiler cannot
    run inside the timed process); (value, source) -- (None, None) when no file of this round exists."""
    for name in ("r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json"):
        path = os.path.join(ROOT, "profiles", name)
        if os.path.exists(path):
            try:
                with open(path) as f:
                    d = json.load(f)
                return d["hbm_bytes_per_launch_avg"], f"profiles/{name} (collected at git {d.get('git', '?')}, {d.get('note', 'FETCH_SIZE + WRITE_SIZE')})"
            except Exception as e:  # a broken file must not pass silently as "no data"
                return None, f"profiles/{name} unreadable: {e!r}"
    return None, None


def masked_conv_flops(plan_cin, plan_cout, k, topo_in, topo_out, allow_same, positions):
    """Algorithmic FLOPs (2 * MAC) of TopoGroupDynamicMaskConv2d (masked_conv.py:102-228) evaluated at `positions` (flat
    ids y * W + x): for every output group the (input group, tap) pairs whose neighbour id is < (<=) the centre id."""
    import numpy as np
    Gi, H, W = topo_in.shape
    Go = topo_out.shape[0]
    half = k // 2
    macs = 0
    ys, xs = np.asarray(positions) // W, np.asarray(positions) % W
    for dy in range(-half, half + 1):
        for dx in range(-half, half + 1):
            ny, nx = ys + dy, xs + dx
            ok = (ny >= 0) & (ny < H) & (nx >= 0) & (nx < W)
            nyc, nxc = np.clip(ny, 0, H - 1), np.clip(nx, 0, W - 1)
            for gi in range(Gi):
                tn = topo_in[gi, nyc, nxc]
                for go in range(Go):
                    tc = topo_out[go, ys, xs]
                    m = ok & ((tn <= tc) if allow_same else (tn < tc))
                    macs += int(m.sum()) * (plan_cin // Gi) * (plan_cout // Go)
    return 2 * macs


def ar_traffic_from_profiles(workload):
    """HBM-side bytes per launch of the AR workloads' dominant kernels from the separate rocprofv3 --pmc passes under profiles/
    (TCC_EA0_RDREQ / WRREQ request counts x request size; scripts/pmc_ar_fold.py): (value, source) or (None, reason)."""
    path = os.path.join(ROOT, "profiles", "r04_pmc_ar.json")
    if not os.path.exists(path):
        return None, "profiles/r04_pmc_ar.json not collected"
    try:
        with open(path) as f:
            d = json.load(f)
        w = d["workloads"][workload]
        return w["hbm_bytes_per_launch"], f"profiles/r04_pmc_ar.json (collected at git {d.get('git', '?')}; {d.get('note', '')})"
    except Exception as e:
        return None, f"profiles/r04_pmc_ar.json unreadable: {e!r}"


def kodak_leg_child(workers):
    """BASELINE configs[3]'s own test setting -- 24 Kodak-shaped images, batch 1 -- through the harness (tools/run_benchmark.py, the
    analogue of the reference's tool) with `workers` stream workers, complexity level 0; a child process."""
    import subprocess
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        cmd = [sys.executable, os.path.join(ROOT, "tools", "run_benchmark.py"), "--warmup", "--codec", "basic", "--synthetic", "24", "--height", "512",
               "--width", "768", "--batch-size", "1", "--workers", str(workers), "--complexity-levels", "0", "--out", os.path.join(tmp, "run")]
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=400, cwd=ROOT)
            if r.returncode != 0:
                return dict(error=f"exit code {r.returncode}", stderr_tail=r.stderr[-300:])
            m = json.loads(r.stdout[r.stdout.index("{"):])
            wall = [v for k, v in m.items() if k.endswith("time_wall_dataset")][0]
            tc = [v for k, v in m.items() if k.endswith("time_compress")][0]
            td = [v for k, v in m.items() if k.endswith("time_decompress")][0]
            return dict(value=24 * 512 * 768 / wall / 1e3, unit="Mpix/s", images=24, shape="3x512x768", batch_size=1, workers=workers, dataset_wall_ms=wall,
                        item_compress_ms=tc, item_decompress_ms=td,
                        note="tools/run_benchmark.py --codec basic --synthetic 24 --height 512 --width 768 --batch-size 1 (synthetic images of Kodak's shape, level 0); "
                             "the harness's timed region: per-item compress() + decompress() incl. upload")
        except Exception as e:
            return dict(error=repr(e))


def run_ar_workload(args):
    """Extra bench lines for the AR parity configurations (BASELINE configs[2] / [3]): whole batches in flight on concurrent stream
    workers as in the headline; after each timed leg the last step's bytes are compared with one quiet single-stream compress();
    the dominant kernels are then timed alone (HIP events per launch) against the algorithmic FLOPs of the reference's masks."""
    # measured: checkerboard 474 / 483 / 492-500 Mpix/s with 4 / 5 / 6 workers (round 3); BaSIC at 64 images per step with the batched
    # persistent scan-line kernel (round 4, profiles/r04_basic_workers.txt): 112 / 153 / 189 / 195 / 200 with 1 / 2 / 3 / 4 / 6
    workers = max(1, args.workers if args.workers is not None else 6)
    if workers > 1:   # before HIP initialises: one hardware queue per stream; 8 image streams per rANS workgroup (CUs left to the others)
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
        os.environ.setdefault("BASIC_RANS_WPB", str(args.rans_waves if args.rans_waves > 0 else 8))
    kodak = None
    if args.workload == "basic" and not args.no_kodak_leg:   # a child process, before this one touches the GPU
        kodak = kodak_leg_child(workers)
    import numpy as np
    import torch
    from cbench_basic_amd.nn import kernels as K
    from cbench_basic_amd.presets import basic_codec, seed_synthetic_weights, topogroup_ar_codec
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    if args.workload == "checkerboard":
        make, batch, name = (lambda: topogroup_ar_codec("checkerboard")), args.batch, "topo-group AR codec, checkerboard + expand-bottleneck merger (lossy_latent_graph_topogroup)"
        levels = [None]
    else:
        make, batch, name = basic_codec, min(args.batch, 64), "BaSIC slimmable scan-line codec (lossy_latent_graph_scalable_ar_models)"
        levels = list(args.levels) if args.levels else [0, 3, 7]
    cpu_state = {}

    def make_codec():
        codec = seed_synthetic_weights(make(), seed=0).eval()
        g = torch.Generator().manual_seed(1)
        with torch.no_grad():
            for n, p in codec.named_parameters():
                if ".latent_node_entropy_coders.y." in n:
                    p.copy_(torch.randn(p.shape, generator=g) * (0.02 if p.dim() > 1 else 0.01))
        if not cpu_state:
            cpu_state.update({k: v.detach().cpu().clone() for k, v in codec.entropy_coder.state_dict().items()})
        codec = codec.to(dev)
        codec.update_state()
        return codec
    # Whole batches in flight on concurrent stream workers, as in the headline: one worker's rANS chains run beside another's
    # convolutions; persistent scan-line launches of different workers sit side by side (64 - 80 compute units each).
    from cbench_basic_amd.benchmark.stream_workers import StreamWorkerPool
    pool = StreamWorkerPool(make_codec, workers, dev)
    codec = pool.codecs[0]
    x = torch.stack([image(i, args.size) for i in range(batch)]).to(dev)

    def loop(c, n):
        out = None
        for _ in range(n):
            data = c.compress(x)
            out = (data, c.decompress(data))
        return out
    counts = [len(range(w, args.steps, workers)) for w in range(workers)]
    per_level = {}
    data = xhat = None
    for lvl in levels:
        if lvl is not None:
            for c in pool.codecs:
                c.set_complex_level(lvl)
        if workers > 1:   # HIP-graph capture / plan building: first call of every replica (at this level) alone
            for c, st in zip(pool.codecs, pool.streams):
                with torch.cuda.stream(st):
                    loop(c, 1)
                torch.cuda.synchronize()
        pool.map(lambda c, n: loop(c, max(1, -(-args.warmup // workers))), counts)
        torch.cuda.synchronize()
        t0 = time.time()
        d_l, xh_l = pool.map(loop, counts)[0]
        torch.cuda.synchronize()
        dt_l = time.time() - t0
        # the timed schedule against ONE quiet call on the idle GPU: same bytes, same reconstruction, or the run fails
        quiet = codec.compress(x)
        xq = codec.decompress(quiet)
        torch.cuda.synchronize()
        same = bool(quiet == d_l and torch.equal(xq, xh_l))
        if not same:
            raise SystemExit(f"bench.py --workload {args.workload}: the bytes / reconstruction of the timed {workers}-worker schedule differ from a quiet single-stream call (level {lvl})")
        mse_l = K.mse_per_image(xh_l, x)
        per_level[lvl] = dict(value=batch * args.size ** 2 * args.steps / dt_l / 1e6, unit="Mpix/s", ms_per_step=dt_l / args.steps * 1e3,
                              bpp=len(d_l) * 8 / (batch * args.size ** 2), psnr_db=float((-10 * torch.log10(mse_l.double())).mean()),
                              bytes_match_single_stream=same)
        if data is None:
            data, xhat, dt = d_l, xh_l, dt_l
    pool.close()
    if levels[0] is not None:
        codec.set_complex_level(levels[0])
    # ---- one instrumented pass (HIP graphs off): events around every masked-convolution launch, FLOPs from the masks
    yc = codec.entropy_coder.latent_node_entropy_coders["y"]
    yc.use_hip_graphs = False
    yc._graphs = {}
    events, flops, per_launch = [], [0], []
    orig = K.MaskedConvPlan.__call__

    def timed(self, xin, topo_in, topo_out, pos, out, out_offset=0, step=None, first_step=None, **layout):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = orig(self, xin, topo_in, topo_out, pos, out, out_offset=out_offset, step=step, first_step=first_step, **layout)
        e1.record()
        events.append((e0, e1))
        ti, to = topo_in.cpu().numpy(), topo_out.cpu().numpy()
        H, W = ti.shape[1:]
        pp = np.unique(pos.cpu().numpy() % (H * W))
        first = first_step.cpu().numpy().reshape(-1) if step is not None else None
        f = 0
        for go in range(to.shape[0]):
            tg = to[go].reshape(-1)
            # the step rule evaluates only the (output group, position) pairs of this step (id-less groups at the first visit)
            q = pp if step is None else [v for v in pp if tg[v] == step or (tg[v] < 0 and first[v] == step)]
            if len(q):
                f += masked_conv_flops(self.cin, self.cout // to.shape[0], self.k, ti, to[go:go + 1], self._same, q)
        flops[0] += f * xin.shape[0]
        per_launch.append((self.cin, self.cout, self.k, int(len(pp)), f * xin.shape[0]))
        return r
    # allow_same is a plan property: remember it at construction for the FLOP count
    yc._layers = None
    orig_init = K.MaskedConvPlan.__init__

    def init(self, weight, bias, in_groups, out_groups, allow_same, act=K.ACT_NONE):
        orig_init(self, weight, bias, in_groups, out_groups, allow_same, act)
        self._same = bool(allow_same)
    K.MaskedConvPlan.__init__, K.MaskedConvPlan.__call__ = init, timed
    yc.use_persistent_scanline = False   # the per-step path: one masked-convolution launch sequence per coding step (counts the algorithmic FLOPs)
    try:
        yc._ready()
        d2 = codec.compress(x)
        codec.decompress(d2)
        torch.cuda.synchronize()
    finally:
        K.MaskedConvPlan.__init__, K.MaskedConvPlan.__call__ = orig_init, orig
    mc_ms = sum(e0.elapsed_time(e1) for e0, e1 in events)
    ach = flops[0] / (mc_ms / 1e3) / 1e12 if mc_ms > 0 else 0.0
    traffic, traffic_source = ar_traffic_from_profiles(args.workload)
    roofline = dict(bound="mfma", achieved=ach, peak=PEAK_FP32_MFMA_TFLOPS, unit="TFLOP/s", frac=ach / PEAK_FP32_MFMA_TFLOPS, traffic=traffic, traffic_source=traffic_source,
                    kernel="masked_conv_dma_kernel / masked_conv_gather_kernel / masked_conv_block_kernel + reduce (csrc/mconv.hip): the masked-convolution launches of one encode + decode pass "
                           "of the y-coder (context convolution + merger layers at the coded positions), HIP events per launch",
                    flops_per_launch=flops[0] / max(1, len(events)), launches_per_pass=len(events), avg_launch_ms=mc_ms / max(1, len(events)),
                    pass_ms=mc_ms, note="algorithmic FLOPs = 2 x the (output, input, tap) products the reference's masks keep at the positions a step codes",
                    launches=None if len(events) > 64 else [
                        dict(cin=ci, cout=co, k=kk, positions_per_image=npos, gflop=fl / 1e9, ms=e0.elapsed_time(e1),
                             tflops=fl / 1e9 / max(e0.elapsed_time(e1), 1e-6)) for (ci, co, kk, npos, fl), (e0, e1) in zip(per_launch, events)])
    if args.workload == "basic":
        # the kernel that serves this workload: ONE persistent launch per direction (scanline_batched_kernel), timed alone with the
        # persistent path switched back on; the per-step figures above stay as the fallback path's
        fallback = dict(achieved=ach, frac=ach / PEAK_FP32_MFMA_TFLOPS, launches_per_pass=len(events), pass_ms=mc_ms, avg_launch_ms=mc_ms / max(1, len(events)))
        yc.use_persistent_scanline = True
        yc._layers = None
        yc._ready()
        sl_events = []
        oenc, odec = K.ScanlinePlan.encode, K.ScanlinePlan.decode

        def tenc(self, *a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = oenc(self, *a, **k)
            e1.record()
            sl_events.append(("encode", e0, e1))
            return r

        def tdec(self, *a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = odec(self, *a, **k)
            e1.record()
            sl_events.append(("decode", e0, e1))
            return r
        K.ScanlinePlan.encode, K.ScanlinePlan.decode = tenc, tdec
        try:
            codec.decompress(codec.compress(x))   # warm
            sl_events.clear()
            d3 = codec.compress(x)
            codec.decompress(d3)
            torch.cuda.synchronize()
        finally:
            K.ScanlinePlan.encode, K.ScanlinePlan.decode = oenc, odec
        if d3 != data or not sl_events:
            raise SystemExit("bench.py --workload basic: the persistent scan-line path did not serve the pass (or coded other bytes than the timed leg)")
        enc_ms = sum(e0.elapsed_time(e1) for w, e0, e1 in sl_events if w == "encode")
        dec_ms = sum(e0.elapsed_time(e1) for w, e0, e1 in sl_events if w == "decode")
        sl_ms = enc_ms + dec_ms
        ach2 = flops[0] / (sl_ms / 1e3) / 1e12
        tiles = -(-batch // 32)
        cus_enc, cus_dec = 32 * tiles, 32 * tiles + -(-batch // 4)
        cus_chip = torch.cuda.get_device_properties(dev).multi_processor_count
        occupied_peak = PEAK_FP32_MFMA_TFLOPS * (enc_ms * cus_enc + dec_ms * cus_dec) / (sl_ms * cus_chip)
        roofline.update(achieved=ach2, frac=ach2 / PEAK_FP32_MFMA_TFLOPS, launches=None,
                        kernel="scanline_batched_kernel<false> + <true> (csrc/scanline.hip): ONE persistent launch per direction walks all H*W coding steps of the batch -- masked context "
                               "convolution + merger layers as v_mfma_f32_32x32x2_f32 tiles with the batch as N, weights resident in registers, tagged-granule exchange between the "
                               "row tiles' compute units, in-kernel rANS decode; HIP events around each launch (incl. its 40 us of prior transpose + memset), alone on the GPU",
                        flops_per_launch=flops[0] / len(sl_events), launches_per_pass=len(sl_events), avg_launch_ms=sl_ms / len(sl_events), pass_ms=sl_ms,
                        encode_launch_ms=enc_ms, decode_launch_ms=dec_ms, compute_units=dict(encode=cus_enc, decode=cus_dec, chip=cus_chip),
                        frac_of_occupied_units=ach2 / occupied_peak,
                        note="algorithmic FLOPs = 2 x the (output, input, tap) products the reference's masks keep at the positions a step codes; `frac` is against the WHOLE chip's "
                             "fp32 MFMA peak although a launch occupies compute_units of it (other workers' launches and transforms run beside it); frac_of_occupied_units = against the "
                             "peak of the units the launches hold.  The decode launch is bound by its serial in-kernel rANS chain (192 symbols per step and stream)",
                        per_step_fallback=fallback)
    first = per_level[levels[0]]
    cfg_levels = None if levels[0] is None else {str(l): v for l, v in per_level.items()}
    out = dict(metric="encode+decode Mpix/s", value=first["value"], unit="Mpix/s", n_gpus=1, steps=args.steps, warmup=args.warmup,
               ms_per_step=first["ms_per_step"], higher_is_better=True, scaling="weak", vs_baseline=None, dtype="f32", data="synthetic",
               config=dict(workload=f"{name}{'' if levels[0] is None else f', complexity level {levels[0]}'}, synthetic 3x{args.size}x{args.size} images, {batch} images per step resident in HBM, "
                                        + (f"{workers} concurrent stream workers (step k on worker k mod W: whole batches in flight)" if workers > 1 else "one stream"),
                           images_per_gpu=batch, workers=workers, bpp=first["bpp"], psnr_db=first["psnr_db"], bytes_match_single_stream=first["bytes_match_single_stream"]),
               roofline=roofline)
    if cfg_levels is not None:
        out["levels"] = cfg_levels
    if kodak is not None:
        out["kodak_batch1"] = kodak
    if not args.no_cpu_baseline:
        from oracle.codec_oracle import BasicCodecOracle, TopoGroupCodecOracle
        from cbench_basic_amd.presets import BASIC_WIDTHS
        try:
            cores = len(os.sched_getaffinity(0))
        except AttributeError:
            cores = os.cpu_count() or 1
        cores = max(1, min(cores, 16))
        torch.set_num_threads(cores)
        oracle = TopoGroupCodecOracle(cpu_state, "checkerboard", 192, 1, True) if args.workload == "checkerboard" else BasicCodecOracle(cpu_state, BASIC_WIDTHS)
        t0, done = time.time(), 0
        for i in range(batch):
            xi = image(i, args.size).unsqueeze(0)
            oracle.decompress(oracle.compress(xi))
            done += 1
            if time.time() - t0 > args.cpu_seconds:
                break
        cdt = time.time() - t0
        out["cpu_baseline"] = dict(value=done * args.size ** 2 / cdt / 1e6, unit="Mpix/s", cores=cores, kind="port",
                                   sample=f"images 0..{done - 1} of the same synthetic set, batch 1, PyTorch-CPU fp32 oracle of the same graph, {cdt:.1f} s wall")
    print(json.dumps(out))


def ar_workload_children(args):
    """The AR parity configurations (BASELINE configs[2] / [3]) as extra keys of the N = 1 line: each runs `bench.py --workload W`
    in a CHILD process started before this process touches the GPU (the workloads set HIP environment variables before HIP
    initialises and instrument the masked-convolution binding), one after the other on the idle GPU; the child's JSON line is
    kept without its per-launch table."""
    import subprocess
    out = {}
    for w in ("checkerboard", "basic"):
        cmd = [sys.executable, os.path.abspath(__file__), "--workload", w, "--steps", str(max(4, args.steps)), "--warmup", "2",
               "--cpu-seconds", "8"] + (["--no-cpu-baseline"] if args.no_cpu_baseline else [])
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
            lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if r.returncode != 0 or not lines:
                out[w] = dict(error=f"exit code {r.returncode}", stderr_tail=r.stderr[-400:])
                continue
            d = json.loads(lines[-1])
            d.get("roofline", {}).pop("launches", None)
            out[w] = {k: d[k] for k in ("value", "unit", "ms_per_step", "steps", "config", "roofline", "levels", "kodak_batch1", "cpu_baseline") if k in d}
        except Exception as e:   # a failed extra line must not take the headline down, but it must be visible
            out[w] = dict(error=repr(e))
    return out


def strong_proxy_child(args):
    """One rank's share of the N = 8 strong-scaling leg (BASELINE configs[4] as written: `total` images per step over 8 GPUs) on
    THIS GPU: `bench.py --batch total/8` with the worker count and token lanes a small batch needs (its ~12 ms of rANS chain
    latency per call against ~3 ms of transforms wants >= 5 batches in flight; its launches leave compute units idle and their
    tile counts quantise badly, so several sessions' transform phases run side by side).  A child process started before this
    one touches the GPU, so the leg runs under exactly the conditions of scripts/workers_sweep.sh (profiles/r03_batch32_*)."""
    import subprocess
    sb, sw, lanes = max(1, args.total // 8), max(2, args.strong_workers), 4
    steps = max(8 * args.steps, 4 * sw)
    cmd = [sys.executable, os.path.abspath(__file__), "--batch", str(sb), "--token-lanes", str(lanes), "--steps", str(steps),
           "--warmup", str(sw), "--size", str(args.size), "--no-cpu-baseline", "--no-extra-legs", "--no-dominant"]   # (no --workers: probed, 4 / 5 / 6)
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode != 0 or not lines:
            return dict(error=f"exit code {r.returncode}", stderr_tail=r.stderr[-400:])
        d = json.loads(lines[-1])
        sw = d["config"].get("workers", sw)
        return dict(value=d["value"], unit="Mpix/s", images_per_step=sb, steps=steps, workers=sw, workers_probe_mpix_s=d["config"].get("workers_probe_mpix_s"),
                    batches_in_flight=sw, token_lanes=lanes, bytes_match_single_stream=d["config"].get("bytes_match_single_stream"),
                    ms_per_step=d["ms_per_step"], call_latency_ms=d["config"].get("call_latency_ms"), predicted_8gpu_strong=8 * d["value"],
                    note=f"BASELINE configs[4] as written is {args.total} images per step over 8 GPUs = {sb} per GPU: this leg runs that share on one GPU "
                         "(image-sharded, no data-path collective, so the N = 8 strong figure is 8 x it up to launch jitter); own process, `bench.py " + " ".join(cmd[2:]) + "`")
    except Exception as e:
        return dict(error=repr(e))


def main():
    args = parse()
    if args.workload != "hyperprior":
        if args.gpus != 1:
            raise SystemExit("the AR workloads are single-GPU extra lines")
        return run_ar_workload(args)
    ar_lines = strong_proxy = None
    if args.gpus == 1 and os.environ.get("WORLD_SIZE") is None and not args.no_extra_legs:   # before anything here initialises the GPU
        if not args.no_ar_workloads:
            ar_lines = ar_workload_children(args)
        if args.batch == args.total:   # the headline configuration: add its strong-scaling share
            strong_proxy = strong_proxy_child(args)
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        raise SystemExit(self_launch(args))
    world = int(env_world or "1")
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus} "
                         f"(or run `python bench.py --gpus {args.gpus}` without a launcher)")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # measured at 256 images per step (scripts/r03_session/r03_call53-55.sh): 3 workers / 1 token lane 671-675 Mpix/s, 4 / 3 676-689,
    # 6 / 4 683-695 (also at 5 timed steps); under-filled tails of one session's launches are filled by another session's
    # The worker count is PROBED at start-up unless --workers names it (never above 6: seven or more stream workers fall off a
    # cliff -- 65-83 Mpix/s whatever the batch, profiles/r03_batch32_sweep.txt -- because six workers x two streams already share the
    # eight hardware queues): 4, 5 and 6 workers run a few steps each on this rank's batch and the fastest is kept (config.workers_probe).
    probe_workers = args.workers is None and args.shard_by == "steps"
    workers = max(1, args.workers if args.workers is not None else 6)
    if workers > 1:  # one hardware queue per worker stream (+ its entropy side stream); HIP's default 4 make streams share
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

    import numpy as np
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or "RANK" in os.environ:  # launched by torch.distributed.run: one rank per GPU
        import torch.distributed as dist
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))  # "nccl" IS RCCL on ROCm (xGMI)
        else:
            dist.init_process_group(args.dist_backend)
    dev = torch.device("cuda", local_rank)

    from cbench_basic_amd.benchmark.stream_workers import StreamWorkerPool, split_batch
    from cbench_basic_amd.nn import kernels as K
    from cbench_basic_amd.presets import hyperprior_codec, seed_synthetic_weights
    from cbench_basic_amd.utils.bytes_ops import split_merged_bytes
    from cbench_basic_amd.utils.dist_metrics import gather_per_image, reduce_metric_sums, shard_indices

    by_steps = args.shard_by == "steps" and workers > 1
    # packing more image streams into one rANS workgroup frees compute units for the other workers' transforms at the
    # price of a longer chain (+2 % at 4, +10 % at 8, x2 at 16): with whole batches in flight the chain has slack
    waves = args.rans_waves if args.rans_waves >= 0 else (8 if by_steps else 4 if workers > 1 else 0)
    def lanes_for(w):   # transform phases admitted at a time: measured best per worker count (profiles/r03_batch32_sweep.txt, r03 call 53-55)
        return args.token_lanes if args.token_lanes is not None else (4 if (args.batch < 128 or w >= 5) else 3 if w == 4 else 1)
    token_lanes = lanes_for(workers)

    def make_codec():
        c = seed_synthetic_weights(hyperprior_codec(), seed=0).eval().to(dev)   # every replica: the same seeded weights
        c.update_state()
        c.entropy_coder.fused_rans_waves = waves
        c.entropy_coder.fused_transform_token = token_lanes if workers > 1 else 0
        return c

    run_workers, run_lanes = [workers], [token_lanes]   # what run_leg() uses by default (updated by the start-up probe)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    def run_leg(pool, batch, steps, warmup, workers=None, lanes=None):
        """Exactly K steps (K x the whole batch through compress + decompress) spread over the first `workers` workers of the
        pool; returns (seconds, bytes of one step, [(bytes, xhat)] of one whole batch in image order, mean seconds of one
        compress+decompress call).  lanes: transform phases admitted at a time in this leg (default: the run's setting)."""
        workers = run_workers[0] if workers is None else workers
        for c in pool.codecs:
            c.entropy_coder.fused_transform_token = (run_lanes[0] if lanes is None else lanes) if workers > 1 else 0
        if by_steps:   # step k on worker k mod W: whole batches, W of them in flight
            work = [(batch, len(range(w, steps, workers)), max(1, len(range(w, warmup, workers))) if warmup else 0)
                    for w in range(workers)]
        else:          # every step's batch cut into W shards
            work = [(s, steps, warmup) for s in split_batch(batch, workers)]
        lat = [0.0] * len(work)

        def loop(i, codec, shard, n):
            out, t = None, time.time()
            for _ in range(n):
                data = codec.compress(shard)
                out = (data, codec.decompress(data))
            lat[i] = (time.time() - t) / max(n, 1)
            return out
        idx = list(range(len(work)))
        pool.map(lambda c, i: loop(i, c, work[i][0], work[i][2]), idx)
        barrier()
        t0 = time.time()
        last = pool.map(lambda c, i: loop(i, c, work[i][0], work[i][1]), idx)
        barrier()
        dt = time.time() - t0
        last = [l for l in last if l is not None]
        if by_steps:
            last = last[:1]
        return dt, sum(len(d) for d, _ in last), last, sum(lat) / len(lat)

    # ---- the rank's images.  weak: ids rank*B .. rank*B+B-1;  strong: image i of `total` lives on rank i mod world
    weak_ids = list(range(rank * args.batch, (rank + 1) * args.batch))
    x_host = torch.stack([image(i, args.size) for i in weak_ids]).pin_memory()
    x = x_host.to(dev)
    # one pool for every leg: the strong-scaling leg (world > 1, or --strong-in-process at N = 1) codes small batches and wants
    # more of them in flight than the headline leg (see strong_proxy_child)
    strong_w = max(workers, args.strong_workers) if (by_steps and (world > 1 or args.strong_in_process)) else workers
    pool = StreamWorkerPool(make_codec, strong_w, dev)
    codec = pool.codecs[0]
    cpu_state = {k: v.detach().cpu().clone() for k, v in codec.entropy_coder.state_dict().items()}

    def strong_leg():
        ids = shard_indices(args.total, rank, world) if world > 1 else list(range(max(1, args.total // 8)))
        xs = torch.stack([image(i, args.size) for i in ids]).to(dev)
        small = len(ids) < 128   # small batches: more of them in flight, four transform phases side by side
        ssteps = args.steps * (8 if small else 1)
        sw, sprobe = (strong_w, None) if not small else (pick_workers(xs, lambda w: 4) if args.workers is None else (strong_w, None))
        sdt, _, _, _ = run_leg(pool, xs, ssteps, sw if small else 1, workers=sw if small else workers, lanes=4 if small else None)
        sred = reduce_metric_sums(dict(time_s=sdt, images=float(len(ids) * ssteps)), device=dev)
        return dict(value=sred["images"] * args.size ** 2 / sred["time_s"] / 1e6, unit="Mpix/s", scaling="strong",
                    images_total_per_step=args.total, images_per_gpu=len(ids), steps=ssteps,
                    workers=sw if small else workers, workers_probe_mpix_s=sprobe, token_lanes=4 if small else token_lanes,
                    ms_per_step=sred["time_s"] / ssteps * 1e3,
                    note="image i of the step's set on rank i mod world; compare with the N=1 line's value (256 images on one GPU)")

    # The strong leg runs BEFORE the weak legs: measured at N = 1 (--strong-in-process), the 32-image leg reaches 523 Mpix/s as
    # the first leg of a process (= the figure of its own process, strong_proxy_child) and after plain 256-image legs
    # (scripts/leg_order_probe.py: 524-532), but 245-288 once the PCIe-inclusive leg has run (BENCH_SKIP_PCIE=1 restores 525):
    # that leg makes three sessions create their copy streams, after which the entropy side streams of the three sessions
    # that first run in the strong leg land on hardware queues that already carry busy streams (15 HIP streams on 8 queues).
    # The 256-image legs are not affected by what ran before them (667 vs 673 Mpix/s, run-to-run spread).
    def pick_workers(batch_t, lanes_of):
        """Times 4, 5 and 6 workers on a few steps of `batch_t` and returns (chosen count, {count: Mpix/s}); every rank takes the same
        count (the slowest rank's view decides).  Never more than six: see the comment at the top of main()."""
        cand = [w for w in (4, 5, 6) if w <= len(pool.codecs)]
        probe = {}
        for w in cand:
            pdt, _, _, _ = run_leg(pool, batch_t, 2 * w, w, workers=w, lanes=lanes_of(w))
            probe[w] = batch_t.shape[0] * 2 * w * args.size ** 2 / pdt / 1e6
        if dist is not None and world > 1:
            t = torch.tensor([probe[w] for w in cand], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            probe = {w: float(v) for w, v in zip(cand, t.tolist())}
        # counts within 1 % of the best are a tie at this probe's length (2 W steps): the LARGEST of them is kept -- over the
        # round's runs six workers gave 695-699 Mpix/s every time, five 684-700 (profiles/README.md)
        best = max(probe.values())
        return max(w for w in cand if probe[w] >= 0.99 * best), probe

    workers_probe = None
    if probe_workers and workers > 1:
        workers, workers_probe = pick_workers(x if args.input == "hbm" else x_host, lanes_for)
        token_lanes = lanes_for(workers)
        run_workers[0], run_lanes[0] = workers, token_lanes
    strong_first = args.strong_first or world > 1
    strong_first_result = None
    if (world > 1 or args.strong_in_process) and strong_first and not args.no_extra_legs:
        strong_first_result = strong_leg()
    dt, step_bytes, last, call_s = run_leg(pool, x if args.input == "hbm" else x_host, args.steps, args.warmup)

    # per-image (bytes, PSNR) of the last step, gathered over xGMI into image order: not part of the timed region
    xhat = torch.cat([xh for _, xh in last])
    psnr_img = -10 * torch.log10(K.mse_per_image(xhat, x).double())
    img_bytes = []
    for data, _ in last:
        zb, yb = split_merged_bytes(data, num_segments=2)
        (_, zo, _), (_, yo, _) = K.unframe_streams(zb), K.unframe_streams(yb)
        img_bytes.append(4.0 * (np.diff(zo) + np.diff(yo)) + 8.0)
    img_bytes = torch.from_numpy(np.concatenate(img_bytes)).to(dev)
    table = torch.stack([img_bytes.double(), psnr_img.to(dev)], dim=1)
    if dist is not None and world > 1:   # contiguous shards per rank -> concatenation in rank order is image order
        bufs = [torch.empty_like(table) for _ in range(world)]
        dist.all_gather(bufs, table)
        table = torch.cat(bufs)
    # the ONLY collective of a leg: metric sums (analogue of cbench/utils/logging_utils.py:458-465)
    red = reduce_metric_sums(dict(time_s=dt, images=float(args.batch * args.steps), bytes=float(step_bytes * args.steps),
                                  psnr_sum=float(psnr_img.sum()), psnr_n=float(args.batch)), device=dev)

    extra = {}
    if not args.no_extra_legs:
        # (1) input in page-locked HOST memory, uploaded inside compress() (the reference's timed region)
        if not os.environ.get("BENCH_SKIP_PCIE"):
            hdt, _, _, _ = run_leg(pool, x_host, args.steps, 1)
            hred = reduce_metric_sums(dict(time_s=hdt, images=float(args.batch * args.steps)), device=dev)
            extra["pcie_inclusive"] = dict(value=hred["images"] * args.size ** 2 / hred["time_s"] / 1e6, unit="Mpix/s",
                                           ms_per_step=hred["time_s"] / args.steps * 1e3,
                                           note="same run, batch in page-locked host memory, H2D inside compress() (general_codec.py:46-47)")
        # (2) BASELINE configs[4] as written: `total` images per step over all GPUs (strong scaling)
        if (world > 1 or args.strong_in_process) and not strong_first:
            extra["strong"] = strong_leg()
    if strong_first_result is not None:
        extra["strong"] = strong_first_result
    pool.close()

    # ---- the timed schedule against ONE quiet call on this rank's idle GPU (no workers, no token, default rANS packing): the last
    # step's bytes and reconstruction must be identical, or the run fails -- the overlapped schedule is what was timed
    bytes_match = None
    if by_steps and args.input == "hbm":
        codec.entropy_coder.fused_transform_token = 0
        codec.entropy_coder.fused_rans_waves = 0
        quiet = codec.compress(x)
        xq = codec.decompress(quiet)
        torch.cuda.synchronize()
        bytes_match = bool(quiet == last[0][0] and torch.equal(xq, last[0][1]))
        if not bytes_match:
            raise SystemExit(f"bench.py: rank {rank}: the bytes / reconstruction of the timed {workers}-worker schedule differ from a quiet single-stream call")
        codec.entropy_coder.fused_rans_waves = waves

    if rank == 0:
        dt_max, n_img, n_bytes = red["time_s"], red["images"], red["bytes"]
        pix = n_img * args.size * args.size
        enc_f, dec_f = conv_flops_per_image(codec, args.size)
        codec.entropy_coder.fused_transform_token = False
        conv_s, launches = measure_conv_kernels(codec, x)
        flops_pass = (enc_f + dec_f) * args.batch
        achieved = flops_pass / conv_s / 1e12
        ms_step = dt_max / args.steps * 1e3
        e2e = flops_pass / (ms_step / 1e3) / 1e12
        traffic, traffic_source = traffic_from_profiles()
        out = dict(
            metric="encode+decode Mpix/s", value=pix / dt_max / 1e6, unit="Mpix/s", n_gpus=world, steps=args.steps,
            warmup=args.warmup, ms_per_step=ms_step, higher_is_better=True, scaling="weak",
            vs_baseline=None, dtype="f32", data="synthetic",
            config=dict(workload=f"hyperprior codec N=128 M=192 (lossy_graph_scalable_exp_hp), synthetic 3x{args.size}x{args.size} images "
                                 f"(image i = manual_seed(i), rand), {args.batch} images per GPU per step "
                                 + ("resident in HBM, " if args.input == "hbm" else "in page-locked HOST memory (uploaded inside compress()), ")
                                 + 
                                 f"compress+decompress incl. bitstream D2H/H2D, {workers} concurrent stream workers per GPU "
                                 + ("(step k on worker k mod W: whole batches, W in flight)" if by_steps else "(each step's batch cut into W shards)"),
                        images_per_gpu=args.batch, workers=workers, workers_probe_mpix_s=workers_probe, token_lanes=token_lanes if workers > 1 else 0,
                        rans_waves_per_workgroup=waves,
                        shard_by=args.shard_by if workers > 1 else None, batches_in_flight=workers if by_steps else 1,
                        call_latency_ms=call_s * 1e3, bytes_match_single_stream=bytes_match,
                        bpp=n_bytes * 8 / pix, psnr_db=red["psnr_sum"] / red["psnr_n"],
                        gathered_images=int(table.shape[0]),
                        gathered_bpp=float(table[:, 0].sum()) * 8 / (table.shape[0] * args.size * args.size),
                        gathered_psnr_db=float(table[:, 1].mean()),
                        parallelism=f"image-sharded x{world}, RCCL all-reduce of metric sums + all-gather of per-image (bytes, PSNR)"),
            roofline=dict(bound="mfma", achieved=achieved, peak=PEAK_FP32_MFMA_TFLOPS, unit="TFLOP/s",
                          frac=achieved / PEAK_FP32_MFMA_TFLOPS, traffic=traffic, traffic_source=traffic_source,
                          kernel="conv_tap_mfma_kernel<MT,CK,KH,KW,WAVES> + first/last-layer kernels (the transform launches of one encode+decode pass, alone on the GPU)",
                          flops_per_launch=flops_pass / launches, launches_per_pass=launches,
                          avg_launch_ms=conv_s / launches * 1e3, pass_ms=conv_s * 1e3,
                          end_to_end=dict(achieved=e2e, frac=e2e / PEAK_FP32_MFMA_TFLOPS,
                                          note="transform FLOPs of a step / wall time of a step (rANS chains, hyper path, copies, host included)"),
                          dominant=None if args.no_dominant else measure_dominant_kernel(codec, x)),
        )
        out.update(extra)
        out["value_per_gpu"] = out["value"] / world
        if "strong" in extra and "value" in extra["strong"]:   # BASELINE configs[4] as written, beside the weak figure
            out["strong_value"] = extra["strong"]["value"]
            out["strong_value_per_gpu"] = extra["strong"]["value"] / world
            out["strong_over_weak"] = extra["strong"]["value"] / out["value"]
        if strong_proxy is not None:
            if "value" in strong_proxy:
                strong_proxy["frac_of_value"] = strong_proxy["value"] / out["value"]
            out["strong_per_gpu_proxy"] = strong_proxy
        if ar_lines is not None:
            out["ar_workloads"] = ar_lines
        if not args.no_cpu_baseline and world == 1:  # reported at N = 1 only; at N > 1 the other ranks would just wait for it
            out["cpu_baseline"] = cpu_baseline(cpu_state, args.cpu_images, args.size)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
