#!/usr/bin/env python3
"""bench.py -- SRADSGAN x4 training images/sec on N MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one full training iteration of the reference's batch loop (SRADSGAN/model/sradsgan.py:
829-892): G forward/backward (+VGG perceptual branch, +D), Adam(G), three D forwards, the WGAN-GP
double backward, Adam(D), weight clip -- on a per-GPU batch of 32 synthetic 54x54 -> 216x216 tiles
(BASELINE.json configs[2]; with N>1 ranks, configs[3]: weak scaling, global batch 32*N, RCCL
all-reduce of G and D gradients).  Inputs are resident in HBM before the timed region.
Before the W warm-up steps every rank runs --spinup-steps (default 25, reported as `spinup_steps`) further untimed
steps: a fresh box runs the first ~1 s of GPU work 20-30 % slower, which 3 warm-up steps do not cover; the timed region
is still exactly K steps between two barrier + synchronize pairs.

Rank 0 prints ONE JSON line; besides the contract fields it carries
  roofline     -- the dominant kernel (3x3 64->256 conv of the RAB stack, in the form the step launches it: padded-plane operands in
                  the default split-bf16 arithmetic, cold operand sets in rotation) timed with HIP events on the launch stream:
                  algorithmic FLOPs per launch / average duration, `traffic` = HBM bytes per launch from the committed PMC passes
                  (refused when the kernel sources have changed since), `in_step_*` = the same launches timed inside extra steps;
  roofline_wgrad -- the same for the weight-gradient launch the step runs (two convolutions per launch + reduce + its conversion share);
  exact_fp32_mode -- the same job re-timed over the same K steps with the conv contraction in exact fp32;
  cpu_baseline -- the CPU oracle (oracle/sradsgan_ref.py, a port of the reference step) timed on the
                  host cores of this box on a bounded sample (rank 0, N=1 only).
"""
import argparse
import json
import os
import math as _math
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# the step overlaps two HIP streams (train_step.py); ROCm multiplexes streams onto a few hardware queues and
# RCCL adds its own -- keep enough queues that the two compute streams never share one (measured: sharing
# costs 10 % of the step).  Must be set before the HIP runtime initialises.
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
# multi-process GPU work on this pool needs dmabuf IPC (RCCL otherwise fails with hipIpcGetMemHandle: invalid argument)
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

PER_GPU_BATCH = 32
SCALE, LR_SIDE = 4, 54
NOMINAL_SCLK_MHZ = 2400.0              # the shader clock the guide's dense peaks are quoted at
FP32_MFMA_PEAK_TFLOPS = 157.3          # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
BF16_MFMA_PEAK_TFLOPS = 2500.0         # same guide: v_mfma_f32_32x32x16_bf16, dense (no sparsity)
# conv arithmetic (include/sradsgan_hip.h srhip_set_conv_math): 'bf16x3' spends three bf16 MFMA products per fp32
# multiply-accumulate, so the ceiling for ALGORITHMIC flops is a third of the bf16 peak
MATH_PEAK = {'fp32': (FP32_MFMA_PEAK_TFLOPS, 'f32 MFMA dense'),
             'bf16x3': (BF16_MFMA_PEAK_TFLOPS / 3.0, 'bf16 MFMA dense 2500 TFLOP/s / 3 products per fp32 MAC'),
             'half': (BF16_MFMA_PEAK_TFLOPS, 'fp16 / bf16 MFMA dense, one product per MAC')}
# headline `dtype`: tensors, accumulators and reductions are fp32 in both modes; the label says how the conv products are formed
DTYPE_LABEL = {'fp32': 'f32 (exact fp32 products)', 'bf16x3': 'f32 (split-bf16 products: 3 bf16 MFMA terms per fp32 multiply, fp32 accumulate)',
               'half': 'f16 (one 16-bit MFMA product per multiply: fp16 on activations, bf16 on gradients; fp32 accumulate and tensors)'}
# algorithmic GFLOP per image per training iteration at each scale of BASELINE configs[4] (HR 216 x 216)
GF_PER_IMG_ITER_BY_SCALE = {2: 986.6, 3: 516.1, 4: 362.6, 8: 216.9, 9: 204.8}
GF_PER_IMG_ITER = 362.6                # algorithmic GFLOP per image per training iteration (x4)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=8)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=PER_GPU_BATCH, help='per-GPU batch (default: BASELINE config)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--graph', action='store_true',
                    help='replay the compute part from a captured hipGraph (experimental, see DESIGN.md section 6)')
    ap.add_argument('--no-graph', action='store_true', help='(default) launch every kernel eagerly')
    ap.add_argument('--conv-math', choices=('fp32', 'bf16x3', 'half'), default=None,
                    help="arithmetic of the conv contraction: fp32 MFMA, split-bf16 x3 MFMA with fp32 accumulate, or one "
                         "16-bit product per multiply (fp16 activations / bf16 gradients; BASELINE configs[4]) "
                         "(default: the library default, sradsgan_amd/_hip.py DEFAULT_CONV_MATH; 'half' for --workload chain)")
    ap.add_argument('--no-fp32-line', action='store_true', help='skip the short extra run in exact-fp32 conv arithmetic')
    ap.add_argument('--cpu-iters', type=int, default=3)
    ap.add_argument('--spinup-steps', type=int, default=25,
                    help='untimed device spin-up steps before the W warm-up steps (same count on every rank): a fresh '
                         'box runs the first ~1 s of GPU work 20-30 %% slower (clocks / first touch), which a few '
                         'warm-up steps do not cover; reported in the JSON line')
    ap.add_argument('--cpu-baseline-only', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--workload', choices=['train', 'infer', 'chain', 'srgan', 'sragan', 'edsr'], default='train',
                    help="'infer': generator-only x4 inference + device metrics (BASELINE configs[1]); 'chain': the training step at "
                         "every scale of the chain sweep x2,x3,x4,x8,x9 on HR 216x216 tiles (BASELINE configs[4]); not the headline line")
    ap.add_argument('--scales', default='2,3,4,8,9', help='--workload chain: comma-separated scale factors')
    ap.add_argument('--roofline-only', action='store_true',
                    help='run only the dominant-kernel measurement (profiles/: rocprofv3 --kernel-trace --stats of this)')
    ap.add_argument('--no-sustained', action='store_true',
                    help='skip the 1.2 s power / clock sampling loops of the roofline kernels (profiler runs: keeps the per-dispatch '
                         'statistics to the timed launches)')
    ap.add_argument('--step-only', action='store_true',
                    help='the timed steps and nothing else: no roofline section, no in-step probe steps, no fp32 line, no CPU baseline '
                         '(A/B runs; profiler runs whose per-kernel table must be the step only).  The line then carries no roofline: not the contract line')
    ap.add_argument('--trace-losses', action='store_true', help='print every step\'s losses to stderr (debug)')
    return ap.parse_args()


def build_networks(device, seed):
    """Random-init networks of the reference architecture (weights_init_normal semantics,
    utils/utils.py:97-114: conv W~N(0,.02), b=0; BN W~N(1,.02), b=0; attention gammas stay 0)."""
    import torch
    from sradsgan_amd import model as M
    g = torch.Generator().manual_seed(seed)
    G = M.GeneratorResNet(M.ResGroup, n_residual_blocks=12, n_basic_blocks=3, rla_mode='CA-SA', bla_mode='CA-SA',
                          ga_mode='CA-SA', pool_mode='Avg|Max', upscale_factor=SCALE)
    D, F = M.Discriminator(), M.FeatureExtractor()
    with torch.no_grad():
        for net in (G, D):
            for m in net.modules():
                name = m.__class__.__name__
                if 'Conv2d' in name:
                    m.weight.copy_(torch.randn(m.weight.shape, generator=g) * 0.02)
                    if m.bias is not None:
                        m.bias.zero_()
                elif 'BatchNorm' in name:
                    m.weight.copy_(1.0 + torch.randn(m.weight.shape, generator=g) * 0.02)
                    m.bias.zero_()
        for m in F.modules():                    # VGG stand-in: He-scaled random weights (no pretrained file offline)
            if 'Conv2d' in m.__class__.__name__:
                fan_in = m.weight.shape[1] * 9
                m.weight.copy_(torch.randn(m.weight.shape, generator=g) * (2.0 / fan_in) ** 0.5)
                m.bias.zero_()
    return G.to(device), D.to(device), F.to(device)


class PowerSampler:
    """Board power and shader clock of one GPU from the amdgpu hwmon files (power1_input in uW, freq1_input in Hz), sampled by
    a thread every 20 ms.  The conv kernels of this path run into the board's power cap (1400 W): the shader clock drops
    from 2.4 GHz to ~1.7 GHz under them (profiles/r02_power_clock_probe.txt), so a roofline fraction quoted against the
    nominal-clock peak is bounded by that, and the line says so with numbers."""

    def __init__(self, device_index=0):
        import glob
        import threading
        self.dir = None
        cands = sorted(glob.glob('/sys/class/drm/card*/device/hwmon/hwmon*/power1_input'))
        want = None
        try:
            import torch
            pr = torch.cuda.get_device_properties(device_index)
            want = '%04x:%02x:%02x' % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        except Exception:
            pass
        for c in cands:
            hw = os.path.dirname(c)
            pci = os.path.basename(os.path.realpath(os.path.join(hw, '..', '..')))
            if want is not None and pci.lower().startswith(want):
                self.dir = hw
        if self.dir is None and len(cands) == 1:
            self.dir = os.path.dirname(cands[0])
        self.samples = []
        self._stop = threading.Event()
        self._thread = threading.Thread(target=self._run, daemon=True) if self.dir else None

    def _read(self, name):
        with open(os.path.join(self.dir, name)) as f:
            return float(f.read().strip())

    def _run(self):
        while not self._stop.is_set():
            try:
                self.samples.append((self._read('freq1_input') / 1e6, self._read('power1_input') / 1e6))
            except (OSError, ValueError):
                pass
            self._stop.wait(0.02)

    def __enter__(self):
        if self._thread is not None:
            self._thread.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        if self._thread is not None:
            self._thread.join()

    def summary(self):
        if not self.samples:
            return None
        f = [a for a, _ in self.samples]
        w = [b for _, b in self.samples]
        try:
            cap = self._read('power1_cap') / 1e6
        except (OSError, ValueError):
            cap = None
        return {'sclk_mhz_mean': round(sum(f) / len(f), 0), 'sclk_mhz_min': round(min(f), 0), 'watts_mean': round(sum(w) / len(w), 0),
                'watts_max': round(max(w), 0), 'power_cap_w': cap, 'nominal_sclk_mhz': NOMINAL_SCLK_MHZ, 'samples': len(f)}


def kernel_source_sha16():
    """First 16 hex digits of the sha256 over the kernel sources and the C header, in name order: what a profile was taken on."""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, 'sradsgan_amd', 'csrc', '*.hip')) + glob.glob(os.path.join(ROOT, 'sradsgan_amd', 'csrc', '*.h'))
                   + [os.path.join(ROOT, 'include', 'sradsgan_hip.h')])
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, 'rb').read())
    return h.hexdigest()[:16]


def _time_launches(fn, iters=50):
    import torch
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def _time_isolated(fn, iters=20):
    """Average of per-launch event pairs with the device drained between launches: what a per-dispatch profiler
    (rocprofv3 --kernel-trace) sees.  Back-to-back launches (_time_launches) can be a few percent faster because the
    drain of one launch's non-temporal stores overlaps the start of the next."""
    import torch
    tot = 0.0
    for _ in range(iters):
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        torch.cuda.synchronize()
        tot += s.elapsed_time(e)
    return tot / iters


def time_dominant_kernel(device, batch, sustained=True, with_single=True):
    """HIP-event timing, on the stream they are launched on, of the two kernels that dominate the step at the
    bench shape [batch,64,54,54] (36 RAB blocks): the conv fprop/dgrad kernel on RAB conv1 (3x3, 64->256, +bias
    +LeakyReLU) -> 'roofline', and the wgrad kernel on the same conv -> second return value."""
    import torch
    from sradsgan_amd import ops
    math = ops.get_conv_math()
    peak, peak_name = MATH_PEAK[math]
    x = torch.randn(batch, 64, LR_SIDE, LR_SIDE, device=device).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(batch, 256, LR_SIDE, LR_SIDE, device=device).contiguous(memory_format=torch.channels_last)
    w = torch.nn.Parameter(torch.randn(256, 64, 3, 3, device=device) * 0.02)
    b = torch.randn(256, device=device) * 0.01
    flops = 2.0 * batch * LR_SIDE * LR_SIDE * 256 * 64 * 9
    # HBM bytes per launch from the committed rocprofv3 --pmc passes (same shape, same mode).  The file names the kernel sources
    # it was measured on (`source_sha16`, tools/roofline_summary.py): counters of an older binary are refused, not reported.
    traffic, traffic_note = {}, None
    if batch == PER_GPU_BATCH:
        try:
            tj = json.load(open(os.path.join(ROOT, 'profiles', 'roofline_traffic.json')))
            if tj.get('source_sha16') == kernel_source_sha16():
                traffic = tj.get(math, {})
            else:
                traffic_note = ('profiles/roofline_traffic.json was measured on other kernel sources (%s, now %s): not reported'
                                % (tj.get('source_sha16'), kernel_source_sha16()))
        except (OSError, ValueError):
            traffic = {}
    out = []
    kf = {'fp32': 'fast_conv_dma_kernel<128,128,bias+lrelu,fp32>', 'bf16x3': 'conv_patch_pers_kernel<128,bias+lrelu> (persistent tile walk)',
          'half': 'conv_patch_pers_kernel<128,run-time epilogue,fp16 single product>'}[math]
    kw = {'fp32': 'fast_wgrad_dma_kernel<128,64,fp32>', 'bf16x3': 'wgrad_rowtap_kernel<128,64>',   # (bf16x3: replaced below by the flat kernel where the RAB runs it)
          'half': 'wgrad_rowtap_kernel<128,64,bf16 single product>'}[math]
    # The step launches the weight gradients of consecutive RABs in pairs (srhip_conv2d_wgrad_multi): the
    # dominant weight-gradient launch therefore processes TWO convolutions, and `roofline_wgrad` times exactly that launch
    # (main kernel + its two reduces) with twice the algorithmic FLOPs; the single-convolution launch of rounds 1-2 is timed
    # next to it (`single_conv_launch_ms`, `single_conv_frac`).
    group = max(1, min(4, int(os.environ.get('SRHIP_WGRAD_GROUP', '2')))) if math != 'fp32' else 1
    x2 = torch.randn_like(x)
    dy2 = torch.randn_like(dy)
    gw = [torch.zeros(256, 64, 3, 3, device=device) for _ in range(group)]
    gb = [torch.zeros(256, device=device) for _ in range(group)]
    items = [((x, x2)[i % 2], (dy, dy2)[i % 2], gw[i], gb[i], 1, 1) for i in range(group)]
    single_wgrad = lambda: ops.conv2d_wgrad_raw(x, dy, (256, 64, 3, 3), 1, 1, True)
    wgrad_fn = (lambda: ops.conv2d_wgrad_multi_raw(items)) if group > 1 else single_wgrad
    kw_label = (kw + ' x%d convolutions per launch + %d reduces' % (group, group)) if group > 1 else (kw + ' + reduce')
    wgrad_extra = {}
    wgrad_calls = 1
    if math == 'bf16x3' and group >= 2 and ops.rab_planes_ok(x, w, torch.empty(64, 256, 3, 3, device='meta')):
        group = ops._PP_GROUP                              # round 6: the step launches a ResGroup's THREE RABs per flat-kernel launch (ops._PP_GROUP)
        gw = [torch.zeros(256, 64, 3, 3, device=device) for _ in range(group)]
        gb = [torch.zeros(256, device=device) for _ in range(group)]
        items = [((x, x2)[i % 2], (dy, dy2)[i % 2], gw[i], gb[i], 1, 1) for i in range(group)]
        nlaunch = 36 // group                              # launches of this shape per step (36 RABs)
        # Round 5: inside the RAB the step keeps dy (the 256-channel gradient, written by conv2's dgrad epilogue) as padded
        # split-bf16 planes and launches the pair on the flat 8-wave kernel (csrc/conv_wgrad_flat.hip); the 64-channel operand x is
        # converted by a small pass on the weight-gradient stream.  Timed here: exactly that -- two pp_from_f32 passes of x + the
        # pair launch + its reduce --, with the pair launch alone (operands already planes) next to it.
        # Operands: THREE sets of two convolutions each (750 MB), launched in rotation, so that no launch finds its operands in the
        # 256 MB Infinity Cache (in the step they were written tens of milliseconds earlier): one set alone reads 20 % faster.
        # In the step a RAB's input x arrives as planes from the tail conv of the block (or, round 5's last change, of the GROUP) in front
        # of it (srhip_conv2d_fwd_dual); only the very first RAB, fed by the head conv, converts x with one pp_from_f32 pass on the
        # weight-gradient stream: one pass per 18 pair launches, which `frac` includes (`conversion_pass_ms`, `pair_launch_alone_*` without it).
        sets = []
        for k in range(3):
            xs_ = [x, x2] + [torch.randn_like(x) for _ in range(group - 2)] if k == 0 else [torch.randn_like(x) for _ in range(group)]
            dys_ = [dy, dy2] + [torch.randn_like(dy) for _ in range(group - 2)] if k == 0 else [torch.randn_like(dy) for _ in range(group)]
            sets.append([(ops.pp_from_f32(xs_[i]), ops.pp_from_f32(dys_[i]), gw[i], gb[i]) for i in range(group)])
        ppx_scratch = ops.pp_empty(batch, 64, LR_SIDE, LR_SIDE, device)
        rot = [0]

        def pair_alone():
            rot[0] = (rot[0] + 1) % 3
            ops.conv2d_wgrad_pp_raw(sets[rot[0]])

        def wgrad_pp():                                     # nlaunch grouped launches + one conversion pass = the 36 conv1 weight gradients of one step
            for _ in range(nlaunch):
                pair_alone()
            ops.pp_from_f32(x, out=ppx_scratch)
        wgrad_fn = wgrad_pp
        wgrad_calls = nlaunch                               # grouped launches per call of wgrad_fn
        pitems = sets[0]
        kw_label = ('wgrad_flat8_kernel<dy planes 256 ch, x planes 64 ch> x%d convolutions per launch + reduce, operands cold (three sets in rotation), '
                    '+ 1/%d pp_from_f32 pass of x per launch (the first RAB of the trunk)' % (group, nlaunch))
        if with_single:                                   # (not under the profiler: its per-kernel averages and byte counters then belong to the launch the step runs)
            wgrad_extra['pair_launch_alone_ms'] = round(_time_launches(pair_alone, 300), 4)       # (key names of round 5: "pair" = the grouped launch, `convolutions_per_launch` convolutions)
            wgrad_extra['conversion_pass_ms'] = round(_time_launches(lambda: ops.pp_from_f32(x, out=ppx_scratch), 200), 4)
            wgrad_extra['rowtap_pair_launch_ms'] = round(_time_launches(lambda: ops.conv2d_wgrad_multi_raw(items), 200), 4)
        single_wgrad = lambda: ops.conv2d_wgrad_pp_raw(pitems[:1])
    fprop_fn = lambda: ops.conv2d_fwd_raw(x, w, b, 1, 1, 0.2)
    fprop_extra = {}
    if math == 'bf16x3' and ops.rab_planes_ok(x, w, torch.empty(64, 256, 3, 3, device='meta')):
        # Round 5: the step runs RAB conv1 on padded split-bf16 planes -- x arrives as planes (from the previous block's tail conv, or
        # from one pp_from_f32 pass for the first block of a ResGroup: counted with the weight gradient, which shares it), t = LeakyReLU(conv1 x)
        # leaves as planes.  Timed: exactly that launch (same kernel template, same FLOPs and algorithmic bytes as the fp32-tensor form:
        # a plane row is the 4 bytes per channel the fp32 row has), on three operand sets in rotation (360 MB: the 256 MB Infinity Cache
        # never holds a launch's operands); the fp32-tensor form of rounds 1-4 is timed beside it (`fp32_tensors_launch_ms`).
        # The training step's launch also leaves the LeakyReLU signs of t as 3 MB of sign words (ops._PP_SIGNS: conv2's data gradient reads
        # those instead of t's hi plane) -- that launch is the one timed; the launch without them (inference) beside it.
        want_signs = ops._PP_SIGNS and ops.pp_sign_words(batch, LR_SIDE, LR_SIDE, 256, device) is not None
        fsets = [(ops.pp_from_f32(x if k == 0 else torch.randn_like(x)), ops.pp_empty(batch, 256, LR_SIDE, LR_SIDE, device),
                  ops.pp_sign_words(batch, LR_SIDE, LR_SIDE, 256, device) if want_signs else None) for k in range(3)]
        frot = [0]

        def fprop_pp(signs=True):
            frot[0] = (frot[0] + 1) % 3
            ops.conv2d_fwd_pp_raw(fsets[frot[0]][0], w, b, 0.2, out_pp=fsets[frot[0]][1], signs=fsets[frot[0]][2] if signs else None)
        fprop_fn = fprop_pp
        kf = ('conv_patch_pers_kernel<128,bias+lrelu,x planes -> t planes%s> (persistent tile walk), operands cold (three sets in rotation)'
              % (' + sign words' if want_signs else ''))
        if with_single:
            fprop_extra['fp32_tensors_launch_ms'] = round(_time_launches(lambda: ops.conv2d_fwd_raw(x, w, b, 1, 1, 0.2), 300), 4)
            if want_signs:
                fprop_extra['without_sign_words_launch_ms'] = round(_time_launches(lambda: fprop_pp(False), 300), 4)
    for key, kernel, fn in (
            ('fprop', kf + ': 3x3 64->256 @54x54 fprop (RAB conv1)', fprop_fn),
            ('wgrad', kw_label + ': 3x3 64->256 @54x54 wgrad (RAB conv1)', wgrad_fn)):
        calls = wgrad_calls if key == 'wgrad' else 1        # launches per call of fn (the weight-gradient mix issues three)
        flops = 2.0 * batch * LR_SIDE * LR_SIDE * 256 * 64 * 9 * (group if key == 'wgrad' else 1)
        # Two short HIP-event measurements, both reported (`achieved` itself comes from the sustained loop below; with
        # --no-sustained from the LONGER of these two): back-to-back launches
        # (sustained clocks; but the drain of one launch's non-temporal stores overlaps the next launch: bf16x3 kernels
        # come out 4-6 % short of a per-dispatch profile) and isolated launches (device drained in between = what
        # rocprofv3 --kernel-trace reports per dispatch; but short bursts run at boost clocks: the fp32-MFMA kernels come
        # out 3-7 % short).  The longer one is within a few percent of the committed rocprofv3 stats in both modes.
        b2b = _time_launches(fn) / calls
        iso = _time_isolated(fn) / calls
        ms = max(b2b, iso)
        achieved = flops / (ms * 1e-3) / 1e12
        rec = {'bound': 'mfma', 'kernel': kernel, 'conv_math': math, 'achieved': round(achieved, 2),
               'peak': round(peak, 1), 'unit': 'TFLOP/s', 'frac': round(achieved / peak, 4),
               'traffic': traffic.get(key), 'flops_per_launch': flops, 'avg_launch_ms': round(ms, 4),
               'back_to_back_launch_ms': round(b2b, 4), 'isolated_launch_ms': round(iso, 4),
               'dtype_peak': peak_name}
        if traffic_note:
            rec['traffic_note'] = traffic_note
        # the same launch looped for ~1.2 s with the board's power and shader clock sampled: `peak` assumes the nominal
        # clock, the kernel runs at whatever clock the 1400 W cap leaves (extra keys, not part of frac)
        if not sustained:                                   # profiler runs: keep the per-dispatch statistics to the timed launches
            rec['power'] = None
            out.append(rec)
            continue
        with PowerSampler(device.index or 0) as ps:
            t_end = time.perf_counter() + 1.2
            n, ev_ms, pairs = 0, 0.0, []
            while time.perf_counter() < t_end:                  # HIP events around every batch of 100 launches, on the launch stream
                s_ev, e_ev = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s_ev.record()
                for _ in range(100):
                    fn()
                e_ev.record()
                pairs.append((s_ev, e_ev))
                if len(pairs) >= 8:                             # keep the host at most 800 launches ahead
                    pairs[0][1].synchronize()
                    ev_ms += pairs[0][0].elapsed_time(pairs[0][1])
                    pairs.pop(0)
                n += 100
            torch.cuda.synchronize()
            for s_ev, e_ev in pairs:
                ev_ms += s_ev.elapsed_time(e_ev)
            sus = ev_ms / n / calls
        pw = ps.summary()
        if pw is not None:
            pw['sustained_launch_ms'] = round(sus, 4)
            pw['frac_of_peak_at_sustained_clock'] = round(flops / (sus * 1e-3) / 1e12 / (peak * pw['sclk_mhz_mean'] / NOMINAL_SCLK_MHZ), 4)
        rec['power'] = pw
        # `achieved` / `frac` are quoted on the average over this sustained loop (n launches, HIP events around batches of
        # 100 on the launch stream): it is the steady state of the kernel, and what rocprofv3 --kernel-trace --stats of the same command
        # averages to (profiles/).  The 50-launch burst and the drained-device figures stay in the record: they catch the
        # clock ramping up from idle and read 15 - 25 % longer.
        rec['avg_launch_ms'] = round(sus, 4)
        rec['launches_timed'] = n * calls
        rec['achieved'] = round(flops / (sus * 1e-3) / 1e12, 2)
        rec['frac'] = round(rec['achieved'] / peak, 4)
        if key == 'fprop':
            for k2, v2 in fprop_extra.items():
                rec[k2] = v2
                if k2 == 'fp32_tensors_launch_ms':
                    rec['fp32_tensors_frac'] = round(flops / (v2 * 1e-3) / 1e12 / peak, 4)
        if key == 'wgrad':
            for k2, v2 in wgrad_extra.items():
                rec[k2] = v2
                if k2 == 'pair_launch_alone_ms':
                    rec['pair_launch_alone_frac'] = round(flops / (v2 * 1e-3) / 1e12 / peak, 4)
        if key == 'wgrad' and group > 1 and with_single:         # (not under the profiler: keeps its per-kernel averages to the grouped launch)
            one = _time_launches(single_wgrad, 200)
            rec['convolutions_per_launch'] = group
            rec['single_conv_launch_ms'] = round(one, 4)
            rec['single_conv_frac'] = round(flops / group / (one * 1e-3) / 1e12 / peak, 4)
        out.append(rec)
    return out[0], out[1]


def in_step_probe(step_fn, batch, kind, steps=4, armed=True):
    """What the step gets: `steps` more identical training steps with the library's timing probe armed on ONE conv geometry
    (RAB conv1, 3x3 64 -> 256 @ 54x54 at this batch: kind 1 = every fprop call, 3 = every weight-gradient call), i.e. HIP events
    around those launches on their own launch stream while the step's other streams share the chip (srhip_probe_*,
    include/sradsgan_hip.h).  Run AFTER the timed region so that the events are not part of `value`.  Returns
    (average ms per convolution, calls, convolutions).
    With more than one rank EVERY rank must call this (the steps carry the gradient exchange's collectives, so a rank that
    skipped them would leave its peers' all-reduces unmatched); only the rank with `armed` arms and reads the probe."""
    import ctypes
    import torch
    from sradsgan_amd import _hip
    lib = _hip.lib()
    cap = 1024
    if armed:
        _hip.check(lib.srhip_probe_config(kind, batch, LR_SIDE, LR_SIDE, 64, 256, cap), 'probe_config')
    n = 0
    try:
        for _ in range(steps):
            step_fn()
        torch.cuda.synchronize()
        if armed:
            ms = (ctypes.c_float * cap)()
            units = (ctypes.c_int * cap)()
            n = lib.srhip_probe_read(ms, units, cap)
    finally:
        if armed:
            lib.srhip_probe_config(0, 0, 0, 0, 0, 0, 0)
    if n <= 0:
        return None, 0, 0
    convs = sum(units[i] for i in range(n))
    return sum(ms[i] for i in range(n)) / max(convs, 1), n, convs


def usable_cores():
    """Host cores this process may really use: affinity mask, capped by the cgroup CPU quota (the GPU
    box runs us in a container; os.cpu_count() reports the whole host and oversubscribing it stalls)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us'):
        try:
            txt = open(path).read().split()
            if path.endswith('cpu.max'):
                if txt[0] != 'max':
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                per = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
                if q > 0:
                    n = min(n, max(1, int(q / per + 0.5)))
        except (OSError, ValueError, IndexError):
            pass
    return max(1, min(n, 32))


def cpu_baseline(iters, budget_s=60.0):
    """The oracle's train_step (a port of sradsgan.py:829-892 to stock torch CPU ops) on B=2 tiles."""
    import torch
    from oracle import sradsgan_ref as O
    torch.manual_seed(0)
    cores = usable_cores()
    torch.set_num_threads(cores)
    B = 2
    G = O.GeneratorResNet(O.ResGroup, n_residual_blocks=12, n_basic_blocks=3, upscale_factor=SCALE)
    D, F = O.Discriminator(), O.FeatureExtractor()
    G.apply(O.weights_init_normal), D.apply(O.weights_init_normal)
    oG = torch.optim.Adam(G.parameters(), lr=2e-4, betas=(0.9, 0.999))
    oD = torch.optim.Adam(D.parameters(), lr=2e-4, betas=(0.9, 0.999))
    lr = torch.rand(B, 3, LR_SIDE, LR_SIDE)
    hr = torch.rand(B, 3, LR_SIDE * SCALE, LR_SIDE * SCALE)
    alpha = torch.rand(B, 1, 1, 1)
    t_start = time.perf_counter()
    O.train_step(G, D, F, oG, oD, lr, hr, alpha)              # warm-up (oneDNN primitive creation)
    ts = []
    for _ in range(iters):
        t0 = time.perf_counter()
        O.train_step(G, D, F, oG, oD, lr, hr, alpha)
        ts.append(time.perf_counter() - t0)
        if time.perf_counter() - t_start > budget_s:          # bounded sample: never hold the bench for minutes
            break
    ts.sort()
    med = ts[len(ts) // 2]
    return {'value': round(B / med, 4), 'unit': 'img/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': 'batch 2 x %d iterations (+1 warm-up) of the identical x4 54->216 training step, '
                      'oracle/sradsgan_ref.train_step on torch CPU ops, median' % len(ts)}


def run_inference(args, device):
    """BASELINE configs[1]: SRADSGAN generator-only x4 inference (mfeNew_validate's G forward, sradsgan.py:1305)
    followed by the device-side validation metrics; batch 16 unless --batch is given."""
    import torch
    from sradsgan_amd import ops, validate
    B = 16 if args.batch == PER_GPU_BATCH else args.batch
    G, _, _ = build_networks(device, seed=20240)
    G.eval()
    gen = torch.Generator().manual_seed(77)
    hr = torch.rand(B, 3, LR_SIDE * SCALE, LR_SIDE * SCALE, generator=gen).to(device)
    lr = torch.rand(B, 3, LR_SIDE, LR_SIDE, generator=gen).to(device)
    if args.no_graph:
        run = lambda: validate.evaluate(G, lr, hr, SCALE)
    else:
        run = validate.GraphedEvaluator(G, SCALE)           # launch-bound at this batch size: replay a captured hipGraph
        run = (lambda ev: (lambda: ev(lr, hr)))(run)
    for _ in range(max(0, args.spinup_steps) + args.warmup):    # untimed spin-up + warm-up
        out = run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = run()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    roof = None
    if not args.step_only:
        # the workload's dominant kernel in isolation at THIS batch (RAB conv1 fprop, 36 of the forward's 78 conv launches): at B = 16 a
        # launch has 768 tiles of the 128-wide walk for 768 block slots, the 64-channel convs 384 -- half a wave of blocks
        roof, _ = time_dominant_kernel(device, B, not args.no_sustained, with_single=False)
        roof['whole_job_frac_of_mfma_peak'] = round(B * args.steps / dt * 69.19 / 1e3 / MATH_PEAK[ops.get_conv_math()][0], 4)
    print(json.dumps({'metric': 'generator inference images/sec (54x54 -> 216x216, x4) incl. device PSNR/SSIM/ERGAS',
                      'value': round(B * args.steps / dt, 2), 'unit': 'img/s', 'n_gpus': 1, 'steps': args.steps,
                      'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 3), 'higher_is_better': True,
                      'dtype': 'f32', 'data': 'synthetic',
                      'config': {'workload': 'SRADSGAN generator-only x4 inference, batch %d' % B,
                                 'conv_math': ops.get_conv_math(), 'launch': 'eager' if args.no_graph else 'hipGraph'},
                      'gflop_per_image': 69.19, 'tflops': round(B * args.steps / dt * 69.19 / 1e3, 2),
                      'mean_psnr_vs_random_target': round(float(out['sr']['psnr'].mean()), 4), 'roofline': roof}), flush=True)


def _use_graph(args):
    """--graph / BENCH_GRAPH=1: replay the three-stream compute part of the step from one captured hipGraph."""
    return (args.graph or os.environ.get('BENCH_GRAPH') == '1') and not args.no_graph


def run_chain(args, device):
    """BASELINE configs[4] on one GPU: the full training step at every scale of the chain sweep (HR tile 216 x 216, LR tile
    216 / s, per-GPU batch as given), default arithmetic 'half'.  One JSON line: per-scale img/s and ms/step, `value` = images
    per second over the whole sweep (the same number of steps at every scale)."""
    import torch
    from sradsgan_amd import model as M, ops
    from sradsgan_amd.train_step import TrainStep
    from sradsgan_amd.trainer import weights_init_normal
    if not args.conv_math:
        ops.set_conv_math('half')
    B = args.batch
    scales = [int(v) for v in args.scales.split(',')]
    per, tot_t, tot_img = {}, 0.0, 0
    for i, sc in enumerate(scales):
        torch.manual_seed(20240 + sc)
        G = M.GeneratorResNet(M.ResGroup, n_residual_blocks=12, n_basic_blocks=3, upscale_factor=sc)
        D, Fx = M.Discriminator(), M.FeatureExtractor()
        G.apply(weights_init_normal), D.apply(weights_init_normal)
        with torch.no_grad():
            G.GAB_UP.ca.gamma.fill_(0.5), G.GAB_UP.sa.gamma.fill_(0.5)        # global attention live, as in the parity tests
            for m in Fx.modules():                                            # VGG stand-in: He-scaled random weights
                if 'Conv2d' in m.__class__.__name__:
                    m.weight.normal_(0.0, (2.0 / (m.weight.shape[1] * 9)) ** 0.5)
                    m.bias.zero_()
        for m in (G, D, Fx):
            m.to(device)
        step = TrainStep(G, D, Fx, use_graph=_use_graph(args))
        gen = torch.Generator().manual_seed(1234 + sc)
        side = 216 // sc
        hr = torch.rand(B, 3, side * sc, side * sc, generator=gen).to(device)
        lr = torch.rand(B, 3, side, side, generator=gen).to(device)
        alpha = torch.rand(B, 1, 1, 1, generator=gen).to(device)
        # every scale is a NEW model: its plane pool and allocator footprint need ~10 steps to settle (round 5 gave only the first
        # scale a spin-up: x4 inside the sweep read 588 img/s against 664 alone -- steps that still created and zero-filled buffers)
        for _ in range((max(0, args.spinup_steps) if i == 0 else min(max(0, args.spinup_steps), 12)) + args.warmup):
            out = step(lr, hr, alpha)
        torch.cuda.synchronize()
        peak0 = torch.cuda.max_memory_allocated()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = step(lr, hr, alpha)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        finite = all(_math.isfinite(float(out[k])) for k in ('loss_G', 'loss_D'))
        gf = GF_PER_IMG_ITER_BY_SCALE.get(sc)
        per['x%d' % sc] = {'img_per_s': round(B * args.steps / dt, 2), 'ms_per_step': round(dt / args.steps * 1e3, 2), 'lr_side': side,
                           'step_tflops': round(B * args.steps / dt * gf / 1e3, 1) if gf else None, 'losses_finite': finite,
                           'peak_mem_gb': round(peak0 / 2 ** 30, 1)}
        tot_t, tot_img = tot_t + dt, tot_img + B * args.steps
        del step, G, D, Fx, hr, lr
        if os.environ.get('BENCH_CHAIN_EMPTY_CACHE', '0') == '1':
            torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats()
    math = ops.get_conv_math()
    print(json.dumps({'metric': 'chain-training images/sec over the scale sweep (HR 216x216 tiles)', 'value': round(tot_img / tot_t, 2),
                      'unit': 'img/s', 'n_gpus': 1, 'steps': args.steps, 'warmup': args.warmup,
                      'ms_per_step': round(tot_t / (args.steps * len(scales)) * 1e3, 2), 'higher_is_better': True,
                      'dtype': DTYPE_LABEL[math], 'data': 'synthetic',
                      'config': {'workload': 'SRADSGAN full GAN training step at scales %s, HR 216x216, per-GPU batch %d' % (scales, B),
                                 'conv_math': math, 'launch': 'hipGraph (three streams captured)' if _use_graph(args) else 'eager'},
                      'per_scale': per}), flush=True)


def run_sibling(args, device):
    """SURVEY 8(f) rank 4, N=1 only: `sragan` = one SRAGAN training iteration (TrainStep around the SRAGAN generator);
    `srgan` = one SRGAN training iteration (srgan.py:335-365; 16 residual blocks, D, VGG
    features) on BASELINE's tile shape (x4, LR 54x54 -> HR 216x216, batch 32); `edsr` = one EDSR L1 training iteration of
    BASELINE configs[0]'s network (Net(3, 256, 32, 2), LR 108x108 -> HR 216x216, batch 4) with torch's Adam."""
    import torch
    from sradsgan_amd import ops
    gen = torch.Generator().manual_seed(4321)
    if args.workload == 'srgan':
        from sradsgan_amd.model import srgan as M
        B = args.batch
        torch.manual_seed(20240)
        G, D, Fx = M.GeneratorResNet(3, 3, 16, SCALE), M.Discriminator(), M.FeatureExtractor()
        G.apply(M.weights_init_normal), D.apply(M.weights_init_normal)
        for m in (G, D, Fx):
            m.to(device)
        for p in Fx.parameters():
            p.requires_grad_(False)
        ops.mark_static(Fx)
        oG = torch.optim.Adam(G.parameters(), lr=2e-4, betas=(0.9, 0.999))
        oD = torch.optim.Adam(D.parameters(), lr=2e-4, betas=(0.9, 0.999))
        hr = torch.rand(B, 3, LR_SIDE * SCALE, LR_SIDE * SCALE, generator=gen).to(device)
        lr = torch.rand(B, 3, LR_SIDE, LR_SIDE, generator=gen).to(device)
        run = lambda: M.train_step(G, D, Fx, oG, oD, lr, hr)['loss_G']
        name = 'SRGAN x4 training step (G 16 blocks + D + VGG features, LSGAN), LR 54x54 -> HR 216x216, batch %d' % B
    elif args.workload == 'sragan':
        # SRAGAN's iteration is SRADSGAN's (sragan.py:539-575) around its own generator (12 residual blocks x 5 basic
        # blocks, sragan.py:465-467): the same TrainStep
        from sradsgan_amd.model import sragan as M
        from sradsgan_amd.train_step import TrainStep
        from sradsgan_amd.trainer import weights_init_normal
        B = args.batch
        torch.manual_seed(20240)
        G = M.GeneratorResNet(M.ResidualBlock_Block_WithAttention, n_residual_blocks=12, n_basic_blocks=5,
                              upscale_factor=SCALE)
        D, Fx = M.Discriminator(), M.FeatureExtractor()
        G.apply(weights_init_normal), D.apply(weights_init_normal)
        with torch.no_grad():
            G.ca.gamma.fill_(0.5), G.sa.gamma.fill_(0.5)            # attention paths live, as in the main bench
        for m in (G, D, Fx):
            m.to(device)
        step = TrainStep(G, D, Fx)
        hr = torch.rand(B, 3, LR_SIDE * SCALE, LR_SIDE * SCALE, generator=gen).to(device)
        lr = torch.rand(B, 3, LR_SIDE, LR_SIDE, generator=gen).to(device)
        alpha = torch.rand(B, 1, 1, 1, generator=gen).to(device)
        run = lambda: step(lr, hr, alpha)['loss_G']
        name = ('SRAGAN x4 training step (G 12x5 attention blocks + D + VGG features, WGAN-GP), LR 54x54 -> HR 216x216, '
                'batch %d' % B)
    else:
        from sradsgan_amd.model import edsr as M
        B = 4 if args.batch == PER_GPU_BATCH else args.batch
        torch.manual_seed(20240)
        net = M.Net(3, 256, 32, 2).to(device)
        opt = torch.optim.Adam(net.parameters(), lr=1e-4)
        hr = torch.rand(B, 3, 216, 216, generator=gen).to(device)
        lr = torch.rand(B, 3, 108, 108, generator=gen).to(device)

        def run():
            opt.zero_grad(set_to_none=True)
            loss = (net(lr) - hr).abs().mean()
            loss.backward()
            opt.step()
            return loss.detach()
        name = 'EDSR x2 L1 training step (Net(3,256,32,2), BASELINE configs[0]), LR 108x108 -> HR 216x216, batch %d' % B
    for _ in range(max(0, args.spinup_steps) + args.warmup):    # untimed spin-up + warm-up
        out = run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = run()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({'metric': 'training images/sec (216x216 HR tiles)', 'value': round(B * args.steps / dt, 2),
                      'unit': 'img/s', 'n_gpus': 1, 'steps': args.steps, 'warmup': args.warmup,
                      'ms_per_step': round(dt / args.steps * 1e3, 3), 'higher_is_better': True, 'dtype': 'f32',
                      'data': 'synthetic', 'config': {'workload': name, 'conv_math': ops.get_conv_math()},
                      'last_loss': round(float(out), 6)}), flush=True)


def cpu_baseline_subprocess(iters, timeout_s=240):
    """Runs the CPU leg in a child process (own thread pool, hard wall-clock bound) and returns its dict."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), '--cpu-baseline-only', '--cpu-iters', str(iters)]
    env = dict(os.environ, HIP_VISIBLE_DEVICES='', CUDA_VISIBLE_DEVICES='')
    try:
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, env=env)
        for line in reversed(out.stdout.strip().splitlines()):
            if line.startswith('{'):
                return json.loads(line)
        return {'value': None, 'unit': 'img/s', 'cores': usable_cores(), 'kind': 'port',
                'sample': 'CPU leg failed: ' + (out.stderr.strip().splitlines() or ['no output'])[-1][:200]}
    except subprocess.TimeoutExpired:
        return {'value': None, 'unit': 'img/s', 'cores': usable_cores(), 'kind': 'port',
                'sample': 'CPU leg exceeded its %d s bound on this host' % timeout_s}


def visible_gpu_count():
    """GPUs this process could use, counted WITHOUT opening the GPU driver in THIS process (the launcher parent must never touch
    HIP before it spawns its ranks, and `torch.cuda.device_count()` falls back to hipGetDeviceCount when amdsmi is absent): KFD
    topology nodes with a non-zero simd_count are the GPU agents; a *_VISIBLE_DEVICES list narrows them like the runtime would.
    The sysfs count is an UPPER bound only (a cgroup may expose fewer render nodes than the host's topology lists; ranks then
    fail at set_device and the supervisor reports it).  Where the topology directory is masked or absent the devices are counted
    by a short-lived child process instead (the parent stays GPU-free)."""
    import glob
    n = 0
    nodes = glob.glob('/sys/class/kfd/kfd/topology/nodes/*/properties')
    readable = False
    for path in nodes:
        try:
            with open(path) as f:
                readable = True
                for line in f:
                    parts = line.split()
                    if len(parts) == 2 and parts[0] == 'simd_count' and int(parts[1]) > 0:
                        n += 1
                        break
        except (OSError, ValueError):
            pass
    if not readable:
        n = _child_gpu_count()
    for var in ('ROCR_VISIBLE_DEVICES', 'HIP_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES'):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(',') if x.strip() != '']))
    return n


def _child_gpu_count(timeout_s=300):
    """Device count as the HIP runtime of a CHILD process sees it (sysfs unavailable); 0 when that fails."""
    import subprocess
    try:
        out = subprocess.run([sys.executable, '-c', 'import torch; print(torch.cuda.device_count())'], capture_output=True,
                             text=True, timeout=timeout_s)
        return max(0, int(out.stdout.strip().splitlines()[-1]))
    except (subprocess.SubprocessError, OSError, ValueError, IndexError):
        return 0


def _touch(path):
    with open(path, 'a'):
        os.utime(path, None)


class Heartbeat:
    """Worker side of the per-rank supervisor: touches BENCH_HB_FILE (when set) so that the supervisor can tell a rank that is
    working from one that sits in a collective its peers never joined."""

    def __init__(self):
        self.path = os.environ.get('BENCH_HB_FILE')

    def __call__(self):
        if self.path:
            try:
                _touch(self.path)
            except OSError:
                pass


def supervise_rank(args, worker_cmd=None, coord_dir=None):
    """One of these per rank whenever N > 1 (under torch.distributed.run as well as under self_launch): the process the launcher
    started stays GPU-free (no torch import, nothing is exec'd over an initialised runtime) and runs the real rank as a CHILD.

    Why: the driver's N > 1 run is the first time the overlapped gradient exchange (dp.GradSync.start: RCCL collectives enqueued
    tens of milliseconds ahead of their inputs on a high-priority stream) meets real peers -- no box with two GPUs was ever
    available to the builder.  If that attempt fails (a rank exits non-zero) or stalls (a rank's heartbeat stops), EVERY rank's
    supervisor stops its worker and starts a FRESH one with SRHIP_DP_HOST_SYNC=1 (dp.py: nothing enqueued ahead of its inputs;
    produce, exchange, consume serially -- costs the overlap, about 1 ms per step) on a new rendezvous port.  The JSON line says
    which mode produced it (`exchange_mode`, `launch_attempt`).  Rank 0's stdout is held back until every rank of the attempt
    has finished, so a failed attempt never leaves a line behind.

    The supervisors agree through files in a directory all ranks derive alike (MASTER_PORT + the launcher's pid):
    fail.<attempt> (any supervisor: this attempt is over), done.<attempt>.<rank>, port.<attempt> (rank 0: rendezvous port of
    a retry), hb.<attempt>.<rank> (the worker's heartbeat)."""
    import socket
    import subprocess
    import tempfile
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    port0 = os.environ.get('MASTER_PORT', '29533')
    if coord_dir is None:
        # all ranks share the launcher as parent: its pid AND its start time (a recycled pid must not inherit a dead job's files)
        try:
            born = open('/proc/%d/stat' % os.getppid()).read().rsplit(')', 1)[1].split()[19]
        except (OSError, IndexError):
            born = '0'
        coord_dir = os.environ.get('BENCH_COORD_DIR') or os.path.join(tempfile.gettempdir(), 'srhip_bench_%s_%d_%s' % (port0, os.getppid(), born))
    os.makedirs(coord_dir, exist_ok=True)
    first_hb_s = float(os.environ.get('BENCH_FIRST_HEARTBEAT_S', '420'))     # fresh box: the first `import torch` alone can take 2 min
    stale_s = float(os.environ.get('BENCH_HEARTBEAT_STALE_S', '150'))
    max_attempts = 1 if os.environ.get('BENCH_NO_RETRY') == '1' else 2
    cmd = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:] if worker_cmd is None else list(worker_cmd)
    path = lambda name: os.path.join(coord_dir, name)

    def leave(code):
        """The last supervisor to leave removes the directory (every rank marks its exit; nobody polls the files after that)."""
        import shutil
        try:
            _touch(path('exit.%d' % rank))
            if all(os.path.exists(path('exit.%d' % r)) for r in range(world)):
                shutil.rmtree(coord_dir, ignore_errors=True)
        except OSError:
            pass
        return code

    rc = 1
    for attempt in range(1, max_attempts + 1):
        env = dict(os.environ, BENCH_WORKER='1', BENCH_ATTEMPT=str(attempt), BENCH_HB_FILE=path('hb.%d.%d' % (attempt, rank)))
        if attempt > 1:
            env['SRHIP_DP_HOST_SYNC'] = '1'
            env['TORCHELASTIC_USE_AGENT_STORE'] = 'False'    # the launcher's store still holds attempt 1's keys: rank 0 hosts a new one
            if rank == 0:
                sock = socket.socket()
                sock.bind(('127.0.0.1', 0))
                new_port = sock.getsockname()[1]
                sock.close()
                with open(path('port.%d.tmp' % attempt), 'w') as f:
                    f.write(str(new_port))
                os.replace(path('port.%d.tmp' % attempt), path('port.%d' % attempt))
            t_wait = time.monotonic() + 60.0
            while not os.path.exists(path('port.%d' % attempt)):
                if time.monotonic() > t_wait:
                    print('bench.py: rank %d never saw the retry port of attempt %d' % (rank, attempt), file=sys.stderr)
                    return leave(1)
                time.sleep(0.05)
            env['MASTER_PORT'] = open(path('port.%d' % attempt)).read().strip()
            env['MASTER_ADDR'] = '127.0.0.1'
        out_path = path('stdout.%d.%d' % (attempt, rank))
        with open(out_path, 'w') as out_f:
            proc = subprocess.Popen(cmd, env=env, stdout=out_f)
        t_start = time.monotonic()
        why = None
        done_written = False
        try:
            while True:
                if os.path.exists(path('fail.%d' % attempt)):
                    why = why or 'a peer reported failure'
                    break
                code = proc.poll()
                if code is not None and not done_written:
                    if code != 0:
                        why = 'worker exited with code %d' % code
                        rc = code
                        _touch(path('fail.%d' % attempt))
                        break
                    _touch(path('done.%d.%d' % (attempt, rank)))
                    done_written = True
                if done_written:
                    if all(os.path.exists(path('done.%d.%d' % (attempt, r))) for r in range(world)):
                        break
                else:
                    try:
                        age = time.time() - os.stat(env['BENCH_HB_FILE']).st_mtime
                        limit = stale_s
                    except OSError:
                        age, limit = time.monotonic() - t_start, first_hb_s
                    if age > limit:
                        why = 'no heartbeat from the worker for %.0f s' % age
                        rc = 124
                        _touch(path('fail.%d' % attempt))
                        break
                time.sleep(0.1)
        finally:
            if proc.poll() is None:
                proc.terminate()                              # the exact child we started
                try:
                    proc.wait(timeout=10.0)
                except subprocess.TimeoutExpired:
                    proc.kill()
                    proc.wait()
        if why is None:                                       # every rank of this attempt finished
            if rank == 0:
                sys.stdout.write(open(out_path).read())
                sys.stdout.flush()
            return leave(0)
        print('bench.py: rank %d, attempt %d (%s exchange) ended: %s%s' % (
            rank, attempt, 'host-synchronised' if attempt > 1 else 'overlapped', why,
            '; retrying with SRHIP_DP_HOST_SYNC=1 in fresh processes' if attempt < max_attempts else ''), file=sys.stderr)
        if rc == 0:
            rc = 1
    return leave(rc)


def self_launch(args, script=None, argv=None, visible=None):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves -- one child process per GPU, the same
    environment torch.distributed.run would give them.  This process never touches the GPU (it does not even import torch:
    devices are counted through sysfs, children are new processes, nothing is exec'd over an initialised runtime).  Rank 0's
    stdout (the JSON line) is passed through.  `script` / `argv` / `visible` exist for tests/test_bench_launch_cpu.py, which
    drives this function with stub rank scripts."""
    import socket
    import subprocess
    have = visible_gpu_count() if visible is None else visible
    if have < args.gpus:
        print('bench.py: --gpus %d but only %d GPU(s) are visible on this node' % (args.gpus, have), file=sys.stderr)
        return 2
    script = os.path.abspath(__file__) if script is None else script
    argv = sys.argv[1:] if argv is None else argv
    sock = socket.socket()
    sock.bind(('127.0.0.1', 0))
    port = sock.getsockname()[1]
    sock.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, script] + list(argv), env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    # Watch ALL ranks (torchrun's behaviour): when one dies -- OOM, ncclCommInitRank failure, the WORLD_SIZE check -- the others
    # would sit in the store rendezvous or in an RCCL collective forever, so the first non-zero exit ends the job; an overall
    # deadline bounds a hang that kills nobody.
    rc = 0
    deadline = time.monotonic() + float(os.environ.get('BENCH_LAUNCH_DEADLINE_S', '1500'))
    try:
        live = list(procs)
        while live:
            for p in list(live):
                code = p.poll()
                if code is None:
                    continue
                live.remove(p)
                if code != 0 and rc == 0:
                    rc = code
                    print('bench.py: rank %d exited with code %d; stopping the other ranks' % (procs.index(p), code), file=sys.stderr)
            if rc != 0:
                break
            if live and time.monotonic() > deadline:
                print('bench.py: ranks still running at the launch deadline; stopping them', file=sys.stderr)
                rc = 124
                break
            if live:
                time.sleep(0.2)
    finally:
        for p in procs:
            if p.poll() is None:
                p.terminate()                             # the exact children we started, never a pattern
        t_kill = time.monotonic() + 10.0
        for p in procs:
            try:
                p.wait(timeout=max(0.1, t_kill - time.monotonic()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    return rc


def main():
    args = parse()
    if args.cpu_baseline_only:
        print(json.dumps(cpu_baseline(args.cpu_iters)), flush=True)
        return
    if args.gpus > 1 and 'RANK' not in os.environ:
        sys.exit(self_launch(args))
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        # a line with n_gpus != --gpus would void the scaling record: refuse instead of running a different job
        raise SystemExit('bench.py: WORLD_SIZE=%d but --gpus %d; run `python bench.py --gpus %d` (it starts the ranks itself) '
                         'or torch.distributed.run --nproc-per-node %d' % (world, args.gpus, args.gpus, args.gpus))
    if args.gpus > 1 and os.environ.get('BENCH_WORKER') != '1' and args.workload == 'train' and not args.roofline_only:
        sys.exit(supervise_rank(args))                        # this process stays GPU-free; the rank itself is its child
    hb = Heartbeat()            # (no beat before the imports: until the first one the supervisor allows BENCH_FIRST_HEARTBEAT_S --
    import torch                # a fresh box's first `import torch` alone can take 2 min -- instead of the stale-heartbeat limit)
    import torch.distributed as dist
    hb()
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: there is no CPU fallback for the HIP path')
    from sradsgan_amd import _hip
    _hip.lib()                                               # fail loudly if the extension is missing
    from sradsgan_amd import ops
    if args.conv_math:
        ops.set_conv_math(args.conv_math)
    conv_math = ops.get_conv_math()
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    force_dist = os.environ.get('BENCH_FORCE_DIST') == '1'      # debug: RCCL path with a single rank
    if world > 1 or force_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        # no device_id: the communicator (and RCCL's own streams) is then created at the first collective,
        # after the step's compute streams exist and own their hardware queues
        dist.init_process_group('nccl', rank=rank, world_size=world)

    if args.roofline_only:
        r0, r1 = time_dominant_kernel(device, args.batch, not args.no_sustained, with_single=False)
        print(json.dumps({'roofline': r0, 'roofline_wgrad': r1}), flush=True)
        return
    if args.workload == 'infer':
        run_inference(args, device)
        return
    if args.workload == 'chain':
        run_chain(args, device)
        return
    if args.workload in ('srgan', 'sragan', 'edsr'):
        run_sibling(args, device)
        return
    from sradsgan_amd.train_step import TrainStep
    from sradsgan_amd import dp
    B = int(os.environ.get('BENCH_BATCH', args.batch))
    G, D, F = build_networks(device, seed=20240)             # identical initial replicas on every rank
    sync = dp.GradSync(world, force=force_dist) if (world > 1 or force_dist) else None
    step = TrainStep(G, D, F, grad_sync=sync, use_graph=_use_graph(args),
                     use_gp=os.environ.get('BENCH_NO_GP') != '1', lr=float(os.environ.get('BENCH_LR', '2e-4')))   # BENCH_LR: diagnostic (0 freezes the weights)
    gen = torch.Generator().manual_seed(1234 + rank)         # disjoint synthetic shards per rank
    hr = torch.rand(B, 3, LR_SIDE * SCALE, LR_SIDE * SCALE, generator=gen).to(device)
    lr = torch.rand(B, 3, LR_SIDE, LR_SIDE, generator=gen).to(device)
    alpha = torch.rand(B, 1, 1, 1, generator=gen).to(device)

    def barrier():
        hb()
        if world > 1 or force_dist:
            # drain this rank's own queues first: the gradient exchange has its own RCCL communicator (csrc/dp_rccl.hip), and a
            # collective of torch's communicator must not be launched while one of the other is still parked behind an event
            # (two communicators whose kernels start in different orders on different ranks is RCCL's classic deadlock)
            torch.cuda.synchronize()
            dist.barrier()
        torch.cuda.synchronize()
        hb()

    if os.environ.get('BENCH_MAIN_STREAM') in ('1', '2'):       # experiment: the step on a non-default (non-blocking) stream; 2: at high priority
        torch.cuda.set_stream(torch.cuda.Stream(priority=-1 if os.environ['BENCH_MAIN_STREAM'] == '2' else 0))
    trace = []
    for _ in range(max(0, args.spinup_steps)):                  # untimed, before the contract's W warm-up steps
        step(lr, hr, alpha)
        hb()
    torch.cuda.synchronize()
    for _ in range(args.warmup):
        out = step(lr, hr, alpha)
        if args.trace_losses:
            trace.append({k: out[k].clone() for k in ('loss_G', 'loss_D')})
        if os.environ.get('BENCH_SYNC_EACH') == '1':
            torch.cuda.synchronize()
    if os.environ.get('BENCH_NO_WARM_BARRIER') != '1':
        barrier()
    sampler = PowerSampler(local_rank)                           # a thread reading two sysfs files every 20 ms
    sampler.__enter__()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step(lr, hr, alpha)
        if args.trace_losses:
            trace.append({k: out[k].clone() for k in ('loss_G', 'loss_D')})
        if os.environ.get('BENCH_SYNC_EACH') == '1':
            torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    sampler.__exit__()
    if trace:
        print('losses per step:', ' '.join('%d:%.4g/%.4g' % (i, float(t['loss_G']), float(t['loss_D']))
                                           for i, t in enumerate(trace)), file=sys.stderr)
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    losses = {k: float(out[k]) for k in ('loss_G', 'loss_D', 'pixel', 'content', 'loss_gan', 'gp')}
    finite = all(_math.isfinite(v) for v in losses.values())

    # The same job with the conv contraction in exact fp32 (a short extra run after the timed region, every rank takes
    # part): reported next to the headline so both arithmetic modes are in one line
    alt = None
    if conv_math == 'bf16x3' and not args.no_fp32_line and not args.step_only:
        ops.set_conv_math('fp32')
        for _ in range(max(2, args.warmup)):
            step(lr, hr, alpha)
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):                         # the same K steps as the headline
            step(lr, hr, alpha)
        barrier()
        dt1 = time.perf_counter() - t1
        ops.set_conv_math('bf16x3')
        if world > 1:
            t = torch.tensor([dt1], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt1 = float(t.item())
        alt = {'conv_math': 'fp32', 'dtype': 'f32 (exact fp32 products)', 'value': round(world * B * args.steps / dt1, 3),
               'ms_per_step': round(dt1 / args.steps * 1e3, 3), 'steps': args.steps,
               'step_frac_of_mfma_peak': round(world * B * args.steps / dt1 * GF_PER_IMG_ITER / 1e3 / (FP32_MFMA_PEAK_TFLOPS * world), 4)}

    # The two roofline kernels INSIDE the step (three streams on a power-capped chip): 2 x 4 more steps after the timed region with
    # the library's probe armed on rank 0.  EVERY rank runs these steps -- with N > 1 each step carries the RCCL all-reduces of the
    # gradient exchange, and a rank that sat in the final barrier instead would leave rank 0's collectives unmatched for ever.
    probes = {}
    if not _use_graph(args) and B == PER_GPU_BATCH and not args.step_only:
        for key, kind in (('roofline', 1), ('roofline_wgrad', 3)):
            probes[key] = (kind,) + tuple(in_step_probe(lambda: step(lr, hr, alpha), B, kind, armed=(rank == 0)))
        barrier()

    if rank == 0:
        ms = dt / args.steps * 1e3
        value = world * B * args.steps / dt
        line = {
            'metric': 'training images/sec (216x216, x4)', 'value': round(value, 3), 'unit': 'img/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': DTYPE_LABEL[conv_math], 'data': 'synthetic',
            'spinup_steps': max(0, args.spinup_steps),
            'config': {'workload': 'SRADSGAN full GAN x4 training step (G+D+VGG perceptual, WGAN-GP), '
                                   'LR 54x54 -> HR 216x216, per-GPU batch %d' % B,
                       'global_batch': world * B, 'parallelism': 'dp%d' % world,
                       'launch': 'hipGraph (three streams captured)' if _use_graph(args) else 'eager'},
            'losses_finite': finite, 'last_losses': {k: round(v, 6) for k, v in losses.items()},
            'step_tflops': round(value * GF_PER_IMG_ITER / 1e3, 2),
            'step_frac_of_mfma_peak': round(value * GF_PER_IMG_ITER / 1e3 / (MATH_PEAK[conv_math][0] * world), 4),
        }
        line['config']['conv_math'] = conv_math
        line['power'] = sampler.summary()                       # board power / shader clock over the timed region (rank 0's GPU)
        if sync is not None:
            line['rccl_ranks'] = sync.rccl_ranks()             # size of the communicator the gradients really went through
            if line['rccl_ranks'] != world:
                raise SystemExit('bench.py: the gradients went through a communicator of %d ranks, the job has %d' % (line['rccl_ranks'], world))
            line['exchange'] = ('srhip_dp_allreduce_bucket on a dedicated HIP stream; G arena in %d parts in reverse layer order from inside its '
                                'backward, D arena behind the D stream' % len({p for t, p, _, _ in sync.parts if t == 'G'}) if sync.parts
                                else 'srhip_dp_allreduce_bucket on a dedicated HIP stream; each arena in one piece after its backward')
            line['exchange_issued_by'] = 'enqueue thread' if sync._enqueuer is not None else 'calling thread'
            line['exchange_mode'] = ('host-synchronised (SRHIP_DP_HOST_SYNC=1: produce, exchange, consume serially)' if sync.host_sync
                                     else 'overlapped (collectives enqueued ahead of their inputs, ordered by events)')
            line['launch_attempt'] = int(os.environ.get('BENCH_ATTEMPT', '1'))
        if alt is not None:
            line['exact_fp32_mode'] = alt
        if args.step_only:
            line['roofline'] = line['roofline_wgrad'] = None
            line['note'] = '--step-only: no roofline section in this run'
        else:
            line['roofline'], line['roofline_wgrad'] = time_dominant_kernel(device, B, not args.no_sustained)
        peak = MATH_PEAK[conv_math][0]
        conv_flops = 2.0 * B * LR_SIDE * LR_SIDE * 256 * 64 * 9
        for key, (kind, ms_conv, calls, convs) in probes.items():
            if ms_conv:
                line[key]['in_step_avg_launch_ms'] = round(ms_conv * (convs / calls), 4)     # per CALL, like avg_launch_ms
                line[key]['in_step_ms_per_convolution'] = round(ms_conv, 4)
                line[key]['in_step_frac'] = round(conv_flops / (ms_conv * 1e-3) / 1e12 / peak, 4)
                line[key]['in_step_calls_timed'] = calls
                line[key]['in_step_note'] = ('HIP events around every %s call of this geometry on its launch stream during 4 extra training steps '
                                             'after the timed region (srhip_probe_*): the other two streams of the step share the chip'
                                             % ('fprop' if kind == 1 else 'weight-gradient'))
        if world == 1 and not args.no_cpu_baseline and not args.step_only:
            line['cpu_baseline'] = cpu_baseline_subprocess(args.cpu_iters)
    if world > 1 or force_dist:
        dist.barrier()
        sync.close()
        dist.destroy_process_group()
    if rank == 0:
        # RCCL prints its version banner through C stdio: flush that first so the JSON line is the last line
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        print(json.dumps(line), flush=True)


if __name__ == '__main__':
    main()
