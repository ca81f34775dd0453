#!/usr/bin/env python
"""bench.py — 1080p P-frames/s of the MI355X-native TDVC encode/reconstruct path.

Contract: `python bench.py --gpus N --steps K --warmup W`.  For N > 1 it runs one rank per GPU under
torch.distributed.run; started WITHOUT a launcher (`python bench.py --gpus 8`) it starts the ranks itself as fresh
child processes before touching the GPU.  A *step* is one P-frame through `VideoCompressor.forward` (eval mode: both
coders' analysis + entropy-model rate + synthesis, motion compensation, fusion, in-loop filter) at 1088x1920 (1080p
padded to x64), batch 1, coded in GOP order with the reference-list rule of tools/predict.py:55-62 (closed loop on the
GPU's own reconstructions).  "Encode" is the reference's own notion (tools/predict.py never passes `is_compress`): the
forward pass with ESTIMATED bits; the real range-coded encode / decode (`VideoCompressor.encode / decode`) is timed by
tools/time_codec.py and reported in DESIGN.md.  Inputs are resident in HBM before the timed region.  Ranks code
independent GOPs (GOP sharding, no collective on the data path): weak scaling.

Prints ONE JSON line on rank 0 (see DESIGN.md §6 for every field).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_PX = 7_664_102            # 2 x 3 832 051 MAC per padded pixel, one P-frame forward
MFMA_F16_PEAK = 2.5e15             # dense fp16/bf16 MFMA peak, MI355X_MICROARCH.md
HBM_PEAK = 8.0e12                  # HBM3E peak, MI355X_MICROARCH.md (~6.3e12 achievable)
HP, WP = 1088, 1920
PMC_FILE = "profiles/r02_conv3x3_64_64_1080p_pmc.txt"


def self_launch(argv, n):
    """`python bench.py --gpus N` without a launcher (how the driver may call it): start the N ranks as fresh child
    processes under torch.distributed.run BEFORE this process has touched the GPU (a GPU-initialised process is never
    exec'ed or forked), relay their output, return the launcher's exit code."""
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    print("[bench] launching: " + " ".join(cmd), file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def build_model(dev):
    from tdvc_amd.model import VideoCompressor
    from tdvc_amd.synth import fill_parameters
    m = VideoCompressor()
    fill_parameters(m)
    return m.to(dev).eval()


def make_inputs(seed, dev):
    from tdvc_amd.synth import make_gop
    import torch.nn.functional as F
    g = make_gop(seed, 7, 1080, 1920)
    g = F.pad(g, (0, 0, 4, 4))                  # utils.pad(.., 64): 1080 -> 1088, centred
    return g.to(dev)


class GopRunner:
    """codes P-frames 1..6 of a 7-frame GOP in order, then starts over"""

    def __init__(self, model, gop, enabled_amp=True):
        self.m, self.g, self.t, self.refs, self.amp = model, gop, 0, None, enabled_amp

    def step(self):
        from tdvc_amd.synth import ref_list
        if self.t == 0:
            self.refs = [self.g[0:1]]
        self.t += 1
        recon, bpp_res, bpp_mv = self.m(self.g[self.t:self.t + 1], ref_list(self.refs), self.amp)
        self.refs.append(recon)
        if self.t == 6:
            self.t = 0
        return recon, bpp_res, bpp_mv


def cpu_baseline(sample_hw=(HP, WP)):
    """the CPU oracle (fp32 PyTorch restatement) timed on this box's host cores: ONE real 1088x1920 P-frame"""
    from oracle.tdvc_ref import VideoCompressor as Ref
    from tdvc_amd.synth import fill_parameters, make_gop, ref_list
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 16))      # a 1-GPU box's CPU share is 16 cores; more threads only oversubscribe
    torch.set_num_threads(cores)
    ref = Ref().eval()
    fill_parameters(ref)
    h, w = sample_hw
    full = (h, w) == (HP, WP)
    if full:
        import torch.nn.functional as F
        g = F.pad(make_gop(1234, 2, 1080, 1920), (0, 0, 4, 4))
    else:
        g = make_gop(1234, 2, h, w)
    refs = ref_list([g[0:1]])
    with torch.no_grad():
        t0 = time.time()
        ro, bro, bmo = ref(g[1:2], refs, False)
        dt = time.time() - t0
    frac = (h * w) / float(HP * WP)
    cpu_baseline.last = {"full": full, "recon": ro, "bpp": float(bro + bmo), "x": g[1:2]}       # for the `parity` field of the line
    return {"value": round(frac / dt, 5), "unit": "1080p P-frames/s" + ("" if full else " (area-scaled)"), "cores": torch.get_num_threads(),
            "host_cores": os.cpu_count(), "cores_available_to_this_process": avail,
            "kind": "port", "seconds": round(dt, 2),
            "sample": ("1 P-frame forward (cfg-2 frame 1, seed 1234) of the fp32 PyTorch oracle at the full 1088x1920" if full else
                       f"1 P-frame forward of the fp32 PyTorch oracle at {h}x{w} ({frac:.3f} of the 1088x1920 pixels), fps scaled by area")}


def hip_parity_frame(model):
    """the frame `cpu_baseline` codes (cfg-2 frame 1, seed 1234, reference list [I, I, I, I]) on the HIP path, both coder modes:
    -> {mode: (recon on the host, bpp)}"""
    import torch.nn.functional as F
    from tdvc_amd.synth import make_gop, ref_list
    g = F.pad(make_gop(1234, 2, 1080, 1920), (0, 0, 4, 4)).cuda()
    refs = ref_list([g[0:1]])
    out = {}
    with torch.no_grad():
        for mode, amp in (("default", True), ("fp32_islands", False)):
            r, br, bm = model(g[1:2], refs, amp)
            out[mode] = (r.float().cpu(), float(br + bm))
    return out


def parity_record(hip):
    """PSNR / rate of the HIP path minus the CPU oracle's on the frame both coded (north_star: within 0.02 dB / 0.001 bpp)"""
    import math
    o = getattr(cpu_baseline, "last", None)
    if not o or not o["full"] or not hip:
        return None
    crop = lambda t: t[:, :, 4:-4]                      # utils.crop: the 1080 visible rows (tools/predict.py:69-70,87)
    ps = lambda a: 10 * math.log10(1.0 / float(((crop(a) - crop(o["x"])) ** 2).mean()))
    rec = {"oracle": "fp32 CPU oracle (oracle/tdvc_ref), closed-form filler weights, cfg-2 frame 1 at 1088x1920, PSNR on the 1080 visible rows",
           "oracle_psnr": round(ps(o["recon"]), 4), "oracle_bpp": round(o["bpp"], 5), "gates": {"dpsnr": 0.02, "dbpp": 0.001}}
    for mode, (r, b) in hip.items():
        rec[mode] = {"dpsnr": round(ps(r) - ps(o["recon"]), 5), "dbpp": round(b - o["bpp"], 6)}
    rec["dpsnr"], rec["dbpp"], rec["mode"] = rec["default"]["dpsnr"], rec["default"]["dbpp"], "default (fp16-in / fp32-accumulate coders)"
    rec["trained_point"] = ("raw fp32 checkpoint at a trained operating point against the AMP-emulating oracle, one frame and a closed-loop GOP: "
                            "tests/test_model_gpu.py::test_trained_raw_checkpoint_parity_amp_oracle*, ::test_trained_closed_loop_gop; numbers in DESIGN.md section 4")
    return rec


def rank_identity(rank, local, dev):
    """what proves that RCCL saw N different devices: (rank, local device, device name, UUID / PCI id, RCCL version)"""
    pr = torch.cuda.get_device_properties(dev)
    uuid = str(getattr(pr, "uuid", "")) or None
    pci = None
    if hasattr(pr, "pci_bus_id"):
        pci = f"{getattr(pr, 'pci_domain_id', 0):04x}:{pr.pci_bus_id:02x}:{getattr(pr, 'pci_device_id', 0):02x}"
    try:
        ver = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception:                                    # noqa: BLE001
        ver = None
    return {"rank": rank, "local_device": local, "name": pr.name, "uuid": uuid, "pci": pci, "rccl": ver, "host": socket.gethostname()}


def gather_ranks(dist, me, world):
    """all-gather the rank identities (RCCL when there is a process group); rank 0 checks that the devices are distinct"""
    if dist is None:
        return [me]
    got = [None] * world
    dist.all_gather_object(got, me)
    keys = [(g_["host"], g_["uuid"] or g_["pci"] or g_["local_device"]) for g_ in got]
    if len(set(keys)) != world:
        raise SystemExit(f"bench.py: {world} ranks but only {len(set(keys))} distinct devices: {got}")
    return got


# The MFMA conv kernels that carry the frame, each with the layer shape behind most of its launches (its "representative
# launch") and the single-launch tool whose rocprofv3 --pmc passes measure its HBM traffic in this run
REP_LAUNCHES = {
    "conv_pair": dict(kind="pair", cin=64, cout=64, H=HP, W=WP, pmc="profiles/r02_conv_pair_1080p_pmc.txt", match="conv_pair",
                      tool=["one_pair.py", str(HP), str(WP), "6"], what="Res_Block = 2 x (3x3 64->64) fused @1088x1920"),
    "conv_row": dict(kind="conv", cin=128, cout=128, H=HP // 2, W=WP // 2, pmc=None, match="conv_row",
                     tool=["one_conv.py", "128", "128", "3", "1", str(HP // 2), str(WP // 2), "6"], what="3x3 128->128 stride 1 @544x960"),
    "conv_row:128->64": dict(kind="conv", kernel="conv_row", cin=128, cout=64, H=HP, W=WP, pmc=None, match="conv_row",
                             tool=["one_conv.py", "128", "64", "3", "1", str(HP), str(WP), "6"], what="3x3 128->64 stride 1 @1088x1920 (64-column strips)"),
    "conv_mfma_v10": dict(kind="conv", cin=64, cout=216, H=HP, W=WP, pmc=None, match="conv_mfma_v10",
                          tool=["one_conv.py", "64", "216", "3", "1", str(HP), str(WP), "6"], what="3x3 64->216 stride 1 @1088x1920 (conv_offset_mask)"),
}


# HBM-bound operators whose traffic is measured the same way (hbm_rooflines)
HBM_TOOLS = {
    "dcn_fused": dict(match="dcn_lds_kernel", tool=["one_dcn.py", "6", "1.5"]),
}


def pmc_traffic_bytes(pmc_file=PMC_FILE):
    """HBM bytes per representative launch from the committed rocprofv3 --pmc summary of THIS round's kernel (separate
    FETCH_SIZE / WRITE_SIZE passes; gfx950 correction: wide coalesced reads report half -> 2 x FETCH_SIZE).  Not measured
    in the bench run itself (PMC needs the profiler): `traffic_source` names the file; null when it is absent."""
    f = os.path.join(ROOT, pmc_file)
    if not os.path.exists(f):
        return None
    vals = {}
    for ln in open(f):
        parts = ln.split()
        if len(parts) == 2:
            try:
                vals[parts[0]] = float(parts[1])
            except ValueError:
                pass
    if "FETCH_SIZE" not in vals or "WRITE_SIZE" not in vals:
        return None
    return (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0


def measure_traffic(kname):
    """HBM bytes per launch of kernel `kname`'s representative launch (REP_LAUNCHES), MEASURED in this run: two `rocprofv3 --pmc`
    child processes (FETCH_SIZE and WRITE_SIZE need passes of their own: MI355X_MICROARCH.md, rocprofv3 PMC slots) on its
    single-launch tool, started BEFORE this process touches the GPU (a GPU-initialised process starts no child).  gfx950
    correction of the guide: a wide coalesced read reports half its bytes -> 2 x FETCH_SIZE.
    -> (bytes per launch, description) or (None, reason)."""
    import csv
    import glob
    import shutil
    import tempfile
    L = REP_LAUNCHES[kname] if kname in REP_LAUNCHES else HBM_TOOLS[kname]
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None, "rocprofv3 not found"
    vals = {}
    env = dict(os.environ, TMPDIR="/tmp")
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        out = tempfile.mkdtemp(prefix="tdvc_pmc_", dir="/tmp")
        try:
            r = subprocess.run([exe, "--pmc", ctr, "--output-format", "csv", "-d", out, "-o", "r", "--", sys.executable,
                                os.path.join(ROOT, "tools", L["tool"][0])] + L["tool"][1:], cwd="/tmp", env=env,
                               stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=240)
            files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None, f"rocprofv3 --pmc {ctr} failed (rc {r.returncode})"
            v = [float(row["Counter_Value"]) for row in csv.DictReader(open(files[0]))
                 if L["match"] in row["Kernel_Name"] and row["Counter_Name"] == ctr]
            if not v:
                return None, f"no {L['match']} rows in the {ctr} pass"
            vals[ctr] = sum(v) / len(v)
        except Exception as ex:                                  # never lose the headline line to the profiler
            return None, f"{type(ex).__name__}: {ex}"[:200]
        finally:
            shutil.rmtree(out, ignore_errors=True)
    return (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0, (
        f"measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child passes on tools/{' '.join(L['tool'])} before the timed region "
        f"(FETCH_SIZE {vals['FETCH_SIZE']:.0f} KB x 2 [gfx950 wide-read correction] + WRITE_SIZE {vals['WRITE_SIZE']:.0f} KB)")


def _time_launches(fn, reps=20, loops=3):
    """average duration of `reps` back-to-back launches of fn() on torch's current stream (= the launch stream): the median of
    `loops` such loops (like the conv representatives: the first loop after an idle period runs 10-12 % slower, the clock is
    still ramping)"""
    fn()
    torch.cuda.synchronize()
    ms = []
    for _ in range(loops):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1) / reps)
    return sorted(ms)[len(ms) // 2]


def hbm_rooflines(model):
    """The HBM-bound operators north_star names, each timed live at 1088x1920 on random operands: algorithmic bytes
    (SURVEY 8d's per-pixel figures) / average launch time / 8 TB/s."""
    from tdvc_amd import ops
    dev = "cuda"
    out = {}
    P = HP * WP

    def entry(ms, nbytes, what, name=None):
        e = {"what": what, "avg_launch_ms": round(ms, 4), "algorithmic_MB": round(nbytes / 1e6, 1),
             "achieved_GBps": round(nbytes / (ms * 1e-3) / 1e9, 1), "peak_GBps": HBM_PEAK / 1e9, "frac": round(nbytes / (ms * 1e-3) / HBM_PEAK, 4)}
        if name in LIVE_TRAFFIC and LIVE_TRAFFIC[name][0]:          # HBM bytes per launch from the PMC counters of this run
            e["traffic"], e["traffic_source"] = LIVE_TRAFFIC[name]
            e["traffic_over_algorithmic"] = round(LIVE_TRAFFIC[name][0] / nbytes, 3)
        return e

    # (1) SPyNet warp at the finest pyramid level: x2 flow upsample + border-mode bilinear warp + 8-channel concat, one kernel
    r = ops.FM(torch.rand(1, HP, WP, 4, device=dev))
    s_ = ops.FM(torch.rand(1, HP, WP, 4, device=dev))
    flo = ops.FM(torch.randn(1, HP // 2, WP // 2, 2, device=dev))
    up = ops.FM.empty(1, HP, WP, 2, dtype=torch.float32, device=dev)
    cat8 = ops.FM.empty(1, HP, WP, 8, device=dev)
    ms = _time_launches(lambda: ops.spynet_level_input(r, s_, flo, up, cat8))
    out["warp"] = entry(ms, 32.0 * P, "spynet_level_input @1088x1920 (flow_warp + x2 flow upsample + concat; 32 B/px: SURVEY 8d)")
    # (2) fused DCN (motion compensation): 64 in + 216 offset/mask + 64 out fp16 values per pixel = 688 B/px
    x = ops.FM(torch.randn(1, HP, WP, 64, device=dev).half())
    om = ops.FM((torch.randn(1, HP, WP, 216, device=dev) * 1.5).half())
    y = ops.FM.empty(1, HP, WP, 64, device=dev)
    dcn = model.mcnet.dconv
    pc = dcn._pk("w", lambda: ops.pack_conv(dcn.weight, dcn.bias, stride=1, pad=1, ck=8 * dcn.deformable_groups, device=dcn.weight.device))
    ms = _time_launches(lambda: ops.dcn_fused(x, om, pc, y, groups=8, act=ops.ACT_LRELU, slope=0.1, round16=True))
    out["dcn_fused"] = entry(ms, 688.0 * P, "dcn_fused @1088x1920, 8 groups, offsets ~N(0, 1.5 px) (688 B/px fp16: SURVEY 8d)", "dcn_fused")
    # (3) SELayer on a 64-channel full-resolution map: read for the pool, read for the scale, write = 3 x 64 fp16 values per pixel
    se = model.motion_est.attn
    o = ops.FM.empty(1, HP, WP, 64, device=dev)
    ms = _time_launches(lambda: se.run(x, out=o))
    out["se"] = entry(ms, 3.0 * 64 * 2 * P, "SELayer(64) @1088x1920: channel_sum + se_gate + scale_act_res (3 x C values/px: SURVEY 8d)")
    return out


LIVE_TRAFFIC = {}      # kernel name -> (bytes per launch, description), filled before the GPU is touched (main)


def roofline_leg(runner):
    """(1) one extra P-frame with HIP events around every conv launch (torch's current stream IS the launch
    stream) -> per-instantiation table; (2) the dominant kernel's representative launch (REP_LAUNCHES) timed live over
    20 launches, and the same for the two other kernels of the class."""
    from tdvc_amd import ops
    ops.PROFILE = []
    runner.step()
    torch.cuda.synchronize()
    prof, ops.PROFILE = ops.PROFILE, None
    agg = {}
    for r in prof:
        a = agg.setdefault(r["kernel"], dict(ms=0.0, flops=0.0, n=0))
        a["ms"] += r["e0"].elapsed_time(r["e1"])
        a["flops"] += r["flops_real"]
        a["n"] += 1
    tot_ms = sum(v["ms"] for v in agg.values())
    tot_fl = sum(v["flops"] for v in agg.values())

    def rep(kname):
        """time the representative launch of one kernel: 20 back-to-back launches, HIP events on the launch stream"""
        L = REP_LAUNCHES[kname]
        kern = L.get("kernel", kname)            # the kernel whose per-frame totals `frame_kernel` quotes
        x = ops.FM(torch.randn(1, L["H"], L["W"], L["cin"], device="cuda").half())
        P = L["H"] * L["W"]
        if L["kind"] == "pair":
            g = torch.Generator().manual_seed(5)
            w = [(torch.randn(64, 64, 3, 3, generator=g) * 0.04).cuda() for _ in range(2)]
            pp = ops.pack_conv_pair(w[0], torch.zeros(64, device="cuda"), w[1], torch.zeros(64, device="cuda"))
            y = ops.conv_pair(x, pp)
            ran = "conv_pair"
            fn = lambda: ops.conv_pair(x, pp, out=y)
            flop = 2 * 2.0 * P * 64 * 64 * 9
            alg = 2.0 * P * (64 + 64)                      # x in, y out; the intermediate map never leaves the CU
        else:
            pc = ops.pack_conv(torch.randn(L["cout"], L["cin"], 3, 3) * 0.04, torch.zeros(L["cout"]), stride=1, pad=1)
            y = ops.conv(x, pc, act=ops.ACT_RELU)
            ran = ops.L.lib().tdvc_last_conv_kernel().decode()
            fn = lambda: ops.conv(x, pc, out=y, act=ops.ACT_RELU)
            flop = 2.0 * P * L["cout"] * L["cin"] * 9
            alg = 2.0 * P * (L["cin"] + L["cout"])
        _time_launches(fn, 10)                             # the first timed loop of a process runs slow (clock ramp)
        ms = sorted(_time_launches(fn) for _ in range(3))[1]        # median of three loops of 20 back-to-back launches
        ach = flop / (ms * 1e-3) / 1e12
        traffic = pmc_traffic_bytes(L["pmc"]) if L["pmc"] else None
        tsrc = (L["pmc"] + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; not measured in this run)") if traffic else None
        if kname in LIVE_TRAFFIC and LIVE_TRAFFIC[kname][0]:
            traffic, tsrc = LIVE_TRAFFIC[kname]
        return {"kernel": kname, "launch": f"{L['what']} on {ran} ({flop / 1e9:.1f} GFLOP, {alg / 1e6:.1f} MB algorithmic)",
                "achieved": round(ach, 2), "peak": MFMA_F16_PEAK / 1e12, "unit": "TFLOP/s", "frac": round(ach * 1e12 / MFMA_F16_PEAK, 4),
                "avg_launch_ms": round(ms, 4), "traffic": traffic,
                "traffic_source": tsrc,
                "hbm_side": {"algorithmic_GBps": round(alg / (ms * 1e-3) / 1e9, 1), "peak_GBps": 8000.0, "frac": round(alg / (ms * 1e-3) / 8e12, 4)},
                "frame_kernel": {"launches_per_frame": agg[kern]["n"], "ms_per_frame": round(agg[kern]["ms"], 3),
                                 "tflops": round(agg[kern]["flops"] / (agg[kern]["ms"] * 1e-3) / 1e12, 2)} if kern in agg else None}

    # the dominant kernel = most milliseconds per frame among the kernels with a representative launch; the others are
    # reported beside it
    cands = [k for k in REP_LAUNCHES if "kernel" not in REP_LAUNCHES[k] and k in agg]
    name = max(cands, key=lambda k: agg[k]["ms"]) if cands else "conv_pair"
    out = {"bound": "mfma"}
    out.update(rep(name))
    out["other_kernels"] = [rep(k) for k in REP_LAUNCHES if k != name]
    out["all_conv_ms_per_frame"] = round(tot_ms, 3)
    out["all_conv_tflops"] = round(tot_fl / (tot_ms * 1e-3) / 1e12, 2)
    out["by_kernel"] = {k: {"ms": round(v["ms"], 3), "n": v["n"], "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2)}
                        for k, v in sorted(agg.items())}
    return out


def train_measure(B, steps, warmup, rank, gpus, dev, dist, graph=False, scale_update="exact"):
    """BASELINE.json configs[2] (1 GPU) / [3] (N GPUs): one optimisation step (forward, backward incl. both entropy
    models, RCCL gradient mean over ranks, clipping, Adam, aux step, re-packing) on B 256x256 P-frame samples per rank
    (weak scaling: the reference's batch 32 over 8 GPUs).  -> (seconds for `steps` steps, max over ranks; last log)"""
    from tdvc_amd.model import VideoCompressor
    from tdvc_amd.synth import fill_parameters, make_gop, ref_list
    from tdvc_amd.train import TrainStep
    torch.manual_seed(1000 + rank)
    m = VideoCompressor()
    fill_parameters(m)
    m = m.to(dev).train()
    xs, rs = [], []
    for i in range(B):                                   # seeds 1000 + sample index
        g = make_gop(1000 + rank * B + i, 7, 256, 256).to(dev)
        xs.append(g[3:4])
        rs.append(ref_list([g[0:1], g[1:2], g[2:3]]))
    x, refs = torch.cat(xs), torch.cat(rs)
    step = TrainStep(m, train_lambda=2048.0, lr=1e-4, loss_scale=128.0, graph=graph, scale_update=scale_update)
    for _ in range(2 if graph else 0):                   # eager steps that precede the capture (not part of --warmup)
        step(x, refs)
    for _ in range(warmup):
        log = step(x, refs)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        log = step(x, refs)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.barrier()
        dt = float(t)
    return dt, log


def train_record(B, steps, warmup, gpus, dt, log):
    flop_step = 3.0 * FLOP_PER_PX * 256 * 256 * B          # training ~ 3x the forward
    return {"metric": "256x256 P-frame training samples/sec (forward + backward + optimizer step, lambda=2048)",
            "value": round(gpus * B * steps / dt, 3), "unit": "samples/s", "n_gpus": gpus, "steps": steps, "warmup": warmup,
            "ms_per_step": round(1e3 * dt / steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16",
            "data": "synthetic",
            "config": {"workload": f"{B} x 256x256 P-frame samples per rank (septuplet frame 3, refs [I,I,x1,x2]), full fwd/bwd incl. entropy models",
                       "global_batch": B * gpus,
                       "parallelism": f"data-parallel x{gpus} (RCCL gradient mean, 16 MB buckets in gradient-completion order, all-reduced under the backward sweep)",
                       "weights": "closed-form filler (no checkpoint ships)"},
            "whole_step_tflops": round(flop_step * steps / dt / 1e12, 2),
            "rd_loss_last": round(log["rd_loss"], 4), "grad_norm_last": round(log["grad_norm"], 3)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", default="1088x1920", help="HxW of the CPU-oracle frame (default: the real 1088x1920 frame, about a minute)")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 --pmc child passes that measure roofline.traffic")
    ap.add_argument("--no-extras", action="store_true", help="skip the fp32-island, HBM-roofline and training legs of the N=1 line")
    ap.add_argument("--mode", choices=("infer", "train"), default="infer",
                    help="infer (default, the headline metric) | train: BASELINE.json configs[2]/[3], one optimisation step per step")
    ap.add_argument("--scale-update", default="exact", choices=("exact", "lagged"),
                    help="--mode train: when a loss-scale change reaches the host (tdvc_amd.train.TrainStep)")
    ap.add_argument("--graph", action="store_true", help="--mode train: replay the forward + backward as one captured HIP graph")
    ap.add_argument("--train-batch", type=int, default=4, help="samples per rank in --mode train (256x256 P-frames)")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:       # no launcher: start the ranks ourselves, before any GPU call
        sys.exit(self_launch(sys.argv[1:], a.gpus))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        sys.exit(f"bench.py: --gpus {a.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if world == 1 and a.mode == "infer" and not a.no_extras and not a.no_pmc:
        for kname in list(REP_LAUNCHES) + list(HBM_TOOLS):     # child processes: must run before this process initialises the GPU
            t_, src_ = measure_traffic(kname)
            LIVE_TRAFFIC[kname] = (t_, src_)
            print(f"[bench] {kname} traffic: {t_} ({src_})", file=sys.stderr, flush=True)
    dist = None
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    if world > 1 or os.environ.get("TDVC_BENCH_FORCE_DIST") == "1":      # the env flag: a ONE-rank RCCL group (tests exercise this path on one GPU)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    ranks_seen = gather_ranks(dist, rank_identity(rank, local, dev), world)

    def finish_dist():
        if dist:
            dist.barrier()
            dist.destroy_process_group()

    if a.mode == "train":
        dt, log = train_measure(a.train_batch, a.steps, a.warmup, rank, a.gpus, dev, dist, a.graph, a.scale_update)
        finish_dist()
        if rank == 0:
            rec = train_record(a.train_batch, a.steps, a.warmup, a.gpus, dt, log)
            rec["ranks_seen"] = ranks_seen
            print(json.dumps(rec))
        return

    model = build_model(dev)
    # SURVEY §8d seeds: cfg-2 (one GPU) codes the GOP of seed 1234; the GOP-sharded runs give every rank its own GOP
    gop = make_inputs(1234 if world == 1 else 2000 + 100 * rank, dev)
    runner = GopRunner(model, gop)
    for _ in range(a.warmup):
        runner.step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        recon, bpp_res, bpp_mv = runner.step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.barrier()
        dt = float(t)
    torch.cuda.synchronize()
    finish_dist()                      # the other ranks are done: the legs below are rank 0's alone (no rank waits in a barrier)
    if rank != 0:
        return

    fps = a.gpus * a.steps / dt
    roof = roofline_leg(runner)
    line = {
        "metric": "1080p P-frames/sec (encode + reconstruct, lambda=2048 config)", "value": round(fps, 3),
        "unit": "frames/s", "n_gpus": a.gpus, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(1e3 * dt / a.steps, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f16", "data": "synthetic",
        "config": {"workload": "1x 1920x1080 (padded 1088x1920) 7-frame synthetic GOP (seed 1234), batch 1, P-frames coded in GOP order",
                   "gop": 7, "batch": 1, "parallelism": f"gop-sharded x{a.gpus}", "weights": "closed-form filler (no checkpoint ships)",
                   "coders": "fp16-in / fp32-accumulate (default, enabled_amp=True); the fp32-island mode (enabled_amp=False) is `fp32_islands` below",
                   "encode": "VideoCompressor.forward with estimated bits, as tools/predict.py (is_compress never set); real range-coded encode/decode: DESIGN.md §1"},
        "whole_path_tflops": round(FLOP_PER_PX * HP * WP * fps / a.gpus / 1e12, 2),
        "whole_path_frac_of_mfma_peak": round(FLOP_PER_PX * HP * WP * fps / a.gpus / MFMA_F16_PEAK, 4),
        "bpp_last": round(float(bpp_res + bpp_mv), 5),
        "ranks_seen": ranks_seen,
        "roofline": roof,
    }
    hip_par = None
    if a.gpus == 1 and not a.no_extras:
        line["roofline_hbm"] = hbm_rooflines(model)
        if not a.no_cpu_baseline and a.cpu_sample.lower() == f"{HP}x{WP}":
            hip_par = hip_parity_frame(model)             # the frame the CPU oracle codes below, both coder modes
        try:                                           # the same frames with both coders as fp32 islands (pnet.py:33,57)
            if not getattr(model, "fp32_islands_supported", False):
                raise RuntimeError("this build has no fp32-island mode")
            r32 = GopRunner(model, gop, enabled_amp=False)
            for _ in range(2):
                r32.step()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            n32 = max(3, a.steps // 3)
            for _ in range(n32):
                r32.step()
            torch.cuda.synchronize()
            d32 = time.perf_counter() - t1
            line["fp32_islands"] = {"value": round(n32 / d32, 3), "unit": "frames/s", "steps": n32, "ms_per_step": round(1e3 * d32 / n32, 3),
                                    "what": "enabled_amp=False: g_a / h_a / h_s / context / entropy_parameters / g_s of both coders in fp32 (v_mfma_f32_32x32x2_f32)"}
        except Exception as ex:                        # never lose the headline line to an optional leg
            line["fp32_islands"] = {"error": f"{type(ex).__name__}: {ex}"[:300]}
        del model, runner
        torch.cuda.empty_cache()
        tsteps = 15
        tdt, tlog = train_measure(4, tsteps, 3, 0, 1, dev, None)
        line["train_step"] = train_record(4, tsteps, 3, 1, tdt, tlog)
    print("[bench] gpu leg: " + json.dumps(line), file=sys.stderr, flush=True)
    if a.gpus == 1 and not a.no_cpu_baseline:
        h, w = (int(v) for v in a.cpu_sample.lower().split("x"))
        line["cpu_baseline"] = cpu_baseline((h, w))
        par = parity_record(hip_par)
        if par:
            line["parity"] = par
    print(json.dumps(line))


if __name__ == "__main__":
    main()
