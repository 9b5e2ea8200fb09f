"""Host-side callers of the hot path, mirroring the reference's drivers:
`args_config` / `prepare_model` / `run_on_images` of run_on_your_images.py (:54-73, :96-178, :183-203) and the
body of main.test() (main.py:833-911) — argument namespace, checkpoint loading, reflect padding, bicubic
pyramid, the model call, crop / de-normalise / round and PSNR.  No cv2 / skimage / argparse side effects.
"""
import math
import os
from argparse import Namespace

import numpy as np
import torch
import torch.nn.functional as F

from pca_comp import DCTParams
from useful import getmodelconfig

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_WEIGHTS = os.path.join(_HERE, "weights", "fLDRnet_X4K1000FPS_exp1_best_PSNR.npz")


# --test3scales ... --test7scales (main.py:243-268): pyramid depth S_tst and the scales / fractions lists of that length.
# --test3scales leaves the --papermodel values (useful.getmodelconfig: S_tst = 3, four scales).
_TEST_SCALES = {
    3: ([8, 16, 32, 64], [4, 16, 64, 256]),
    4: ([8, 16, 32, 64, 128], [4, 16, 64, 256, 1024]),
    5: ([8, 16, 32, 64, 128, 256], [4, 16, 64, 256, 1024, 4096]),
    6: ([8, 16, 32, 64, 128, 256, 512], [4, 16, 64, 256, 1024, 4096, 16384]),
    7: ([8, 16, 32, 64, 128, 256, 512, 1024], [4, 16, 64, 256, 1024, 4096, 16384, 65536]),
}


def args_config(gpu=0, test_scales=5):
    """The namespace `run_on_your_images.args_config()` produces (--papermodel --test5scales); test_scales = 3 / 4 / 6 / 7
    gives what main.py builds under --papermodel --test<n>scales (main.py:240-273)."""
    from fLDRnet import DCTXVFInet
    a = Namespace(
        gpu=gpu, net_type='fLDRnet', exp_num=1, text_dir='./text_dir', checkpoint_dir='./checkpoint_dir',
        dataset='X4K1000FPS', test5scales=True, parameters=-1, save_images=False,
        softsplat=False, ownsmooth=False, forwendflowloss=False, ownoccl=False, sminterp=False, sminterpWT=False,
        tparam=1, noResidAddup=False, cutoffUnnec=False, fixsmoothtwistup=False, impmasksoftsplat=False,
        TOptimization=False, sminterpInpIm=False, tempAdamfix=False, simpleEVs=False, smallenrefine=False,
        interpOrigForw=False, interpBackwForw=False, inter4k_stepsize=16, noPCA=False, tempbottomflowfix=False,
        pcanet=False, net_object=DCTXVFInet, dctvfi_nf=16, scales=[4, 8, 16, 32, 64, 128],
        fractions=[1, 4, 16, 64, 256, 1024], ref_feat_extrac=False, maskLess=False, imageUpInp=False, allImUp=False,
        ExacOneEV=False, papermodel=True, validation_patch_size=512, meanVecParam=True, align_cornerse=False,
        takeBestModel=True, oneEV=False, optimizeEV=False, noEVOptimization=False, moreTstSc=False,
        padding="reflective", XVFIPSNR=False, continue_training=False, specificCheckpoint=-1, img_ch=3, nf=64,
        S_trn=3, S_tst=5, timetest=False, testgetflowout=False, outMaskLess=False,
    )
    getmodelconfig(a)                                   # run_on_your_images.py:190-191
    if test_scales not in _TEST_SCALES:
        raise ValueError("test_scales must be one of %s (main.py:243-268)" % sorted(_TEST_SCALES))
    a.test5scales = test_scales == 5
    for n in (3, 4, 6, 7):
        setattr(a, "test%dscales" % n, test_scales == n)
    a.scales, a.fractions = (list(v) for v in _TEST_SCALES[test_scales])   # :193-203 / main.py:243-268
    a.moreTstSc = test_scales != 3
    a.phase = "test"
    a.S_tst = test_scales
    a.dctvfi_nf = a.scales[0] ** 2 // a.fractions[0]
    a.padding = "reflect"
    a.takeBestModel = True
    return a


def npz_state_dict(path=DEFAULT_WEIGHTS):
    """Plain-tensor re-export of the shipped checkpoint -> a state dict `load_state_dict(strict=True)` accepts:
    restores the base_modules.* aliases and the unused (all-zero / never-read) entries."""
    z = np.load(path)
    sd = {k: torch.from_numpy(z[k]) for k in z.files}
    for k in list(sd):
        if k.startswith("rec_ctx_ds."):
            sd["base_modules.0." + k[len("rec_ctx_ds."):]] = sd[k]
        elif k.startswith("vfinet."):
            sd["base_modules.1." + k[len("vfinet."):]] = sd[k]
    return sd


def prepare_model(device=None, weights=DEFAULT_WEIGHTS, args=None):
    """run_on_your_images.prepare_model (:54-73) without the save_manager side effects."""
    args = args or args_config()
    device = device or torch.device('cuda:' + str(args.gpu))
    model = args.net_object(args)
    sd = npz_state_dict(weights)
    own = model.state_dict()
    for k, v in own.items():          # entries the export dropped: zero-filled, shape from the module
        if k not in sd:
            sd[k] = torch.zeros_like(v)
    model.load_state_dict(sd, strict=True)
    model.save_params([DCTParams(wiS=8, components_fraction=1 / 4, data_used=0.5) for _ in range(len(args.scales))])
    model.to(device).eval()
    return model, device, args


def pad_frames(frames, args):
    """[B,C,T,H,W] -> reflect-padded to multiples of 2^S_tst*8 on the [B,C*T,H,W] view (main.py:840-849)."""
    B, C, T, H, W = frames.shape
    div = (2 ** args.S_tst) * 8 if args.phase == "test" else (2 ** args.S_trn) * 8
    ph = (div - H % div) % div
    pw = (div - W % div) % div
    x = F.pad(frames.reshape(B, C * T, H, W), (0, pw, 0, ph), args.padding)
    return x.reshape(B, C, T, H + ph, W + pw)


def build_pyramid(frames_padded, args):
    """Direct (non-cascaded) bicubic downscales by 2^-i (main.py:855-856)."""
    B, C, T, H, W = frames_padded.shape
    flat = frames_padded.permute(0, 2, 1, 3, 4).reshape(B * T, C, H, W)
    pyr = [frames_padded]
    for i in range(1, args.S_tst + 1):
        s = args.scales[0] / args.scales[i]
        d = F.interpolate(flat, scale_factor=s, mode='bicubic', align_corners=args.align_cornerse)
        pyr.append(d.reshape(B, T, C, int(H * s), int(W * s)).permute(0, 2, 1, 3, 4).contiguous())
    return pyr


def interpolate(model, args, frames, t_value, pyramid=None):
    """One (pair, t) forward as main.test()/run_on_images do it.  frames [B,3,2,H,W] in [-1,1] on the model's
    device; t_value [B,1].  Returns the fp64 prediction cropped to the original H x W."""
    B, C, T, OH, OW = frames.shape
    with torch.no_grad():
        if pyramid is None:
            pyramid = build_pyramid(pad_frames(frames, args), args)
        dummy = [None] * (args.S_tst + 1)          # the reference passes zero tensors that are overwritten (fLDRnet.py:134)
        pred, _ = model(dummy, t_value, normInput=pyramid, is_training=False, validation=False)
    return pred[:, :, :OH, :OW]


class GraphedInterpolator:
    """`interpolate` captured once in a hipGraph and replayed (opt-in: a serving loop that interpolates frame pairs of ONE shape).

    The forward of a (frames, t) slot is ~61 kernel launches through the C ABI: enqueued eagerly from Python they cost ~1 ms of host
    time per pair, a replay ~0.03 ms — the same kernels in the same order on the same stream, the same bits (checked at capture).
    Usage:
        g = GraphedInterpolator(model, args, frames, t)            # captures on `stream` (default: a stream of its own)
        out = g(frames, t)                                        # copies the inputs into the captured slots, replays, returns the
                                                                  # slot's output tensor (overwritten by the next call)
        g.replay(join=True)                                       # inputs already written in place (g.frames / g.t / g.pyramid)
    `pyramid=` fixes a prebuilt pyramid as the input instead of the frames (bench.py: pyramids resident in HBM).  One instance per
    stream / slot in flight; capture uses a memory pool of its own (or `pool=`, to share one among the slots of a stream).
    check=True: the first replay is compared with an eager forward at once; check="defer": the caller does first_replay() (no
    synchronisation) and verify() later — bench.py replays all its instances back to back right in front of its warm-up that way.
    The pair cache is not captured (model.pair_cache must be off)."""

    def __init__(self, model, args, frames, t_value, pyramid=None, stream=None, pool=None, check=True):
        if model.pair_cache:
            raise RuntimeError("GraphedInterpolator captures the plain forward: switch model.pair_cache off")
        self.model, self.args = model, args
        self.stream = stream if stream is not None else torch.cuda.Stream(device=frames.device)
        self.frames = frames.clone() if pyramid is None else frames
        self.t = t_value.clone()
        self.pyramid = pyramid
        self.stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.stream), torch.no_grad():       # prime the stream's allocator pool and every lazy weight prepack
            ref = interpolate(model, args, self.frames, self.t, pyramid=self.pyramid)
        self.stream.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph, pool=pool, stream=self.stream):
            self.out = interpolate(model, args, self.frames, self.t, pyramid=self.pyramid)
        self._ref = self._first = None
        if check == "defer":                                         # the caller replays first (first_replay) and compares later (verify): no
            self._ref = ref                                          # synchronisation between the capture and the caller's loop
        elif check:                                                  # one replay against the eager frame: the same bits, or no graph
            with torch.cuda.stream(self.stream):
                self.graph.replay()
            self.stream.synchronize()
            if not torch.equal(ref, self.out):
                raise RuntimeError("the replayed frame differs from the eager frame")
        del ref

    def first_replay(self):
        """check="defer": replay once and keep a copy of the frame (on the instance's stream, no synchronisation) for verify()."""
        self.replay()
        with torch.cuda.stream(self.stream):
            self._first = self.out.clone()

    def verify(self):
        """check="defer": the kept first replay against the eager frame of the capture — the same bits, or RuntimeError.  Synchronises."""
        if self._ref is None or self._first is None:
            raise RuntimeError("verify() needs check='defer' and a first_replay()")
        self.stream.synchronize()
        ok = torch.equal(self._ref, self._first)
        self._ref = self._first = None
        if not ok:
            raise RuntimeError("the replayed frame differs from the eager frame")

    def replay(self, join=False):
        """Replay on the instance's stream.  join: the caller's current stream waits for it (device-side) before using the output;
        without it the caller synchronises itself (bench.py keeps several instances in flight and joins once)."""
        import fldr_hip
        fldr_hip.poll_status()                   # the fault flags of earlier replays (no synchronisation), as DCTXVFInet.forward does on entry
        with torch.cuda.stream(self.stream):
            self.graph.replay()
        if join:
            torch.cuda.current_stream().wait_stream(self.stream)
        return self.out

    def __call__(self, frames, t_value):
        """New inputs of the captured shape: ordered behind the caller's stream (which produced them), copied into the captured slots,
        replayed; the caller's stream waits for the result."""
        if self.pyramid is not None:
            raise RuntimeError("captured on a prebuilt pyramid: write the pyramid tensors in place and call replay()")
        self.stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.stream):
            self.frames.copy_(frames, non_blocking=True)
            self.t.copy_(t_value, non_blocking=True)
        return self.replay(join=True)


def interpolate_multi(model, args, frames, t_values, pyramid=None, streams=None):
    """All outputs of one pair (e.g. t = 1/8 ... 7/8 for the 8x X-Test / Inter4K protocol, main.py:833-867) with the
    pair-invariant stage (PCA features, six flow levels, splat metrics) computed once.  Returns a list of frames.
    streams: optional list of torch.cuda.Stream — the first output (which fills the cache) runs on the current stream, the
    others, independent of each other from there on, are dealt round-robin to `streams` and joined before returning."""
    B, C, T, OH, OW = frames.shape
    prev = model.pair_cache
    model.pair_cache = True
    try:
        with torch.no_grad():
            if pyramid is None:
                pyramid = build_pyramid(pad_frames(frames, args), args)
            outs = []

            def one(tv):
                t = torch.full((B, 1), float(tv), device=frames.device, dtype=torch.float32)
                pred, _ = model([None] * (args.S_tst + 1), t, normInput=pyramid, is_training=False, validation=False)
                return pred[:, :, :OH, :OW]
            if not streams or len(t_values) < 2:
                outs = [one(tv) for tv in t_values]
            else:
                cur = torch.cuda.current_stream()
                outs.append(one(t_values[0]))
                ready = torch.cuda.Event()
                ready.record(cur)
                used = []
                try:
                    for i, tv in enumerate(t_values[1:]):
                        st = streams[i % len(streams)]
                        st.wait_event(ready)
                        if st not in used:
                            used.append(st)
                        with torch.cuda.stream(st):
                            outs.append(one(tv))
                finally:
                    # Join the side streams on EVERY path, before the pair cache is dropped below: the cached flow / z0 / z1 and the
                    # pyramid belong to the current stream's allocator pool and kernels already queued on the side streams still read
                    # them if one(tv) raised half-way; the caller's stream owns every output produced so far from here on.
                    for st in used:
                        cur.wait_stream(st)
                    for o in outs[1:]:
                        o.record_stream(cur)
    finally:
        model.pair_cache = prev
        model._pair_state = None
    return outs


U8_DIRECT = True      # interpolate_u8 without a ground truth: the 8-bit frame straight from the synthesis kernel (False: fp64 frame + fldr_frame_metrics)


def interpolate_u8(model, args, frames_u8, t_value, target_u8=None, want_ssim=False):
    """End-to-end on the device: uint8 frames [B,2,3,H,W] (I0, I1) -> normalise + reflect pad + bicubic pyramid
    (fldr_ingest_u8 / fldr_pyramid_bicubic) -> forward -> rounded uint8 frame [B,3,H,W] (and, with a uint8 ground
    truth, the per-sample PSNR list; with want_ssim (PSNR list, SSIM-Y list): main.py:910-911) via fldr_frame_metrics /
    fldr_ssim_y_u8.  Only uint8 crosses PCIe in either direction."""
    import fldr_hip
    B, T, C, H, W = frames_u8.shape
    with torch.no_grad():
        pyr = fldr_hip.ingest_pyramid(frames_u8, args.S_tst + 1)
        # without a ground truth nothing but the rounded frame is wanted: the fused synthesis kernel emits it directly (the fp64 frame is
        # then never written); the model ignores the request on the paths that cannot honour it and returns the fp64 frame as ever
        direct = U8_DIRECT and target_u8 is None and W % 2 == 0 and hasattr(model, "vfinet")
        pred, _ = model([None] * (args.S_tst + 1), t_value, normInput=pyr, is_training=False, validation=False,
                        **({"emit_u8": (H, W)} if direct else {}))       # (an explicit argument of the call: no state on the model)
        if pred.dtype == torch.uint8:
            return pred.contiguous(), None
        sse, img = fldr_hip.frame_metrics(pred, min(H, pred.shape[2]), min(W, pred.shape[3]), target_u8, want_u8=True)
        ssim = fldr_hip.ssim_y_u8(img, target_u8) if (want_ssim and target_u8 is not None) else None
    if sse is None:
        return img, None
    mse = (sse / (3.0 * H * W)).cpu()
    ps = [float("inf") if m == 0 else 10 * math.log10(255.0 ** 2 / m) for m in mse.tolist()]
    return (img, (ps, ssim.cpu().tolist())) if ssim is not None else (img, ps)


def to_uint8_image(pred):
    """[3,H,W] in [-1,1] -> rounded [H,W,3] in [0,255] (main.py:885-894, utils.py:685-688)."""
    p = np.asarray(pred.detach().cpu() if torch.is_tensor(pred) else pred, dtype=np.float64)
    return np.around(((np.transpose(p, [1, 2, 0]) + 1.0) / 2.0).clip(0.0, 1.0) * 255.0)


def psnr(img_true, img_pred):
    """utils.psnr with XVFIPSNR False (utils.py:644-652): data_range 255 over all channels."""
    err = np.mean((np.asarray(img_true, dtype=np.float64) - np.asarray(img_pred, dtype=np.float64)) ** 2)
    return float("inf") if err == 0 else 10 * math.log10(255.0 ** 2 / err)


def ssim_bgr(img_true, img_pred):
    """utils.ssim_bgr (utils.py:662-669) for two [H,W,3] BGR images with rounded values in [0,255] (numpy or tensors),
    computed on the device by fldr_ssim_y_u8."""
    dev = torch.device('cuda', torch.cuda.current_device())
    to = lambda a: torch.as_tensor(np.asarray(a)).round().clamp(0, 255).to(torch.uint8).permute(2, 0, 1).unsqueeze(0).to(dev)
    import fldr_hip
    return float(fldr_hip.ssim_y_u8(to(img_pred), to(img_true))[0].item())


def frames_from_uint8(u8):
    """[2,3,H,W] uint8 (I0, I1) -> [1,3,2,H,W] fp32 in [-1,1] (run_on_your_images.py:84-87)."""
    return ((u8.float() / 255) * 2 - 1).permute(1, 0, 2, 3).unsqueeze(0).contiguous()


def synthetic_pair(H, W, seed=0, quadrant=False, device="cpu"):
    """Seeded synthetic uint8 frame pair (bench.py, tests, golden fixtures): a multi-octave (1/f-like) random
    texture, so that every pyramid level sees structure as in natural video; I1 is I0 shifted by (6,4) px, or by
    (+-12,+-8) px per quadrant to force occlusions/holes.  (5x5-smoothed white noise, the first recipe, has no
    content left below 1/8 resolution and the flow network then predicts meaningless +-40 px flows.)"""
    g = torch.Generator().manual_seed(seed)
    Hb, Wb = H + 64, W + 64
    base = torch.zeros(1, 3, Hb, Wb)
    for o in range(8):
        s = 2 ** o
        n = torch.rand(1, 3, -(-Hb // s) + 2, -(-Wb // s) + 2, generator=g)
        if o:
            n = F.interpolate(n, scale_factor=s, mode="bilinear", align_corners=False)
        base += n[..., :Hb, :Wb] * (1.5 ** o)
    base = (base - base.amin()) / (base.amax() - base.amin())
    I0 = base[..., 32:H + 32, 32:W + 32]
    if not quadrant:
        I1 = base[..., 36:H + 36, 38:W + 38]
    else:
        I1 = I0.clone()
        h2, w2 = H // 2, W // 2
        for (ys, xs, dy, dx) in ((0, 0, 8, 12), (0, 1, -8, 12), (1, 0, 8, -12), (1, 1, -8, -12)):
            y0, x0 = ys * h2, xs * w2
            I1[..., y0:y0 + h2, x0:x0 + w2] = base[..., 32 + y0 + dy:32 + y0 + dy + h2, 32 + x0 + dx:32 + x0 + dx + w2]
    u8 = lambda a: (a.clamp(0, 1) * 255).round().to(torch.uint8)
    return torch.stack([u8(I0[0]), u8(I1[0])], 0).to(device)


def synthetic_pair_varying(H, W, seed=0, device="cpu", zoom=1.012, rot_deg=0.25, shift=(5.0, 3.0)):
    """The same texture under a smoothly VARYING motion: I1 is I0 seen through a 1.2 % zoom about the frame centre plus a
    0.25 degree rotation and a (5, 3) px shift (defaults) — displacements of up to ~30 px at the corners of a 4K frame whose x
    and y components change from pixel to pixel, as camera motion in natural video does; larger zoom / rot_deg for stronger
    non-rigid motion (2.5 % / 0.8 degrees: ~85 px at the corners, sources of one row spread over a dozen target rows).  (A global shift, synthetic_pair's
    default, is the easiest case for the scatter kernels: every row of sources lands on one row of targets.)"""
    import math
    g = torch.Generator().manual_seed(seed)
    Hb, Wb = H + 128, W + 128
    base = torch.zeros(1, 3, Hb, Wb)
    for o in range(8):
        s = 2 ** o
        n = torch.rand(1, 3, -(-Hb // s) + 2, -(-Wb // s) + 2, generator=g)
        if o:
            n = F.interpolate(n, scale_factor=s, mode="bilinear", align_corners=False)
        base += n[..., :Hb, :Wb] * (1.5 ** o)
    base = (base - base.amin()) / (base.amax() - base.amin())
    I0 = base[..., 64:H + 64, 64:W + 64]
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    cy, cx, z, a = (H - 1) / 2, (W - 1) / 2, zoom, math.radians(rot_deg)
    dx, dy = xs - cx, ys - cy
    sx = cx + z * (math.cos(a) * dx - math.sin(a) * dy) + shift[0] + 64
    sy = cy + z * (math.sin(a) * dx + math.cos(a) * dy) + shift[1] + 64
    grid = torch.stack([sx / (Wb - 1) * 2 - 1, sy / (Hb - 1) * 2 - 1], -1).unsqueeze(0)
    I1 = F.grid_sample(base, grid, mode="bilinear", padding_mode="border", align_corners=True)
    u8 = lambda t: (t.clamp(0, 1) * 255).round().to(torch.uint8)
    return torch.stack([u8(I0[0]), u8(I1[0])], 0).to(device)



# ---- dataset-shaped evaluation (X-Test / Inter4K style folders of PNG frames) ----------------------------------------------
def list_xtest_triplets(root, multiple=8, t_step_size=32):
    """The sample list of utils.make_2D_dataset_X_Test (utils.py:414-432) — <root>/<type>/<scene>/*.png, consecutive frames
    t_step_size apart are (I0, I1), the multiple - 1 frames between them the targets at t = k / multiple — GROUPED BY PAIR,
    so that the pair-invariant stage runs once per pair: [(I0_path, I1_path, 'type/scene', [(It_path, t), ...]), ...]."""
    import glob
    ts = np.linspace(1 / multiple, 1 - 1 / multiple, multiple - 1)
    pairs = []
    for type_folder in sorted(glob.glob(os.path.join(root, '*', ''))):
        for scene_folder in sorted(glob.glob(os.path.join(type_folder, '*', ''))):
            frames = sorted(glob.glob(scene_folder + '*.png'))
            for idx in range(0, len(frames), t_step_size):
                if idx == len(frames) - 1 or idx + t_step_size >= len(frames):
                    break
                targets = [(frames[idx + int((t_step_size // multiple) * (m + 1))], float(ts[m])) for m in range(multiple - 1)]
                pairs.append((frames[idx], frames[idx + t_step_size], os.path.relpath(scene_folder, root), targets))
    return pairs


def load_bgr_u8(path):
    """cv2.imread(path) without cv2: uint8 [H,W,3] in BGR order (the checkpoint was trained on cv2's channel order)."""
    from PIL import Image
    with Image.open(path) as im:
        return np.ascontiguousarray(np.asarray(im.convert("RGB"))[:, :, ::-1])


def reduce_sums(values, device="cpu"):
    """SUM-reduce a short list of floats over the ranks (fp64; [sum PSNR, sum SSIM, count]: 24 bytes, the only collective of an
    evaluation run — main.py:960-962 keeps these in AverageClass on its single process)."""
    import torch.distributed as dist
    t = torch.tensor([float(v) for v in values], device=device, dtype=torch.float64)
    if _collectives_on():
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [float(x) for x in t.tolist()]


def evaluate_dir(root, multiple=8, t_step_size=32, model=None, args=None, device=None, want_ssim=True, rank=0, world=1,
                 predict=None, max_pairs=None, log=None):
    """main.test() on a directory (main.py:815-911, utils.py:208-251,644-669): walk X-Test-style folders, interpolate the
    multiple - 1 intermediate frames of every pair with the pair-invariant cache, score each against its ground-truth PNG with
    PSNR (data_range 255, all channels) and SSIM-Y on the DEVICE (fldr_frame_metrics / fldr_ssim_y_u8: only uint8 frames cross
    PCIe), pairs dealt round-robin to the ranks, three sums reduced at the end.
    predict(frames_u8 [1,2,3,H,W] uint8 on `device`, ts, targets_u8 list of [1,3,H,W]) -> [(psnr, ssim or None), ...] replaces the
    model (the CPU test of the walk / sharding / reduction uses it).
    -> dict(psnr, ssim, frames, pairs, per_t={t: psnr}, scenes={scene: psnr}) — the same on every rank."""
    pairs = list_xtest_triplets(root, multiple, t_step_size)
    if not pairs:
        raise RuntimeError("Found 0 files in subfolders of: " + root + "\n")            # utils.py:454-458
    if max_pairs:
        pairs = pairs[:max_pairs]
    if predict is None:
        import fldr_hip
        if model is None:
            model, device, args = prepare_model(device, args=args)
        device = device or next(model.parameters()).device

        def predict(frames_u8, ts, targets_u8):
            B, T, C, H, W = frames_u8.shape
            prev = model.pair_cache
            model.pair_cache = True
            res = []
            try:
                with torch.no_grad():
                    pyr = fldr_hip.ingest_pyramid(frames_u8, args.S_tst + 1)
                    for tv, tgt in zip(ts, targets_u8):
                        t = torch.full((B, 1), float(tv), device=frames_u8.device, dtype=torch.float32)
                        pred, _ = model([None] * (args.S_tst + 1), t, normInput=pyr, is_training=False, validation=False)
                        sse, img = fldr_hip.frame_metrics(pred, min(H, pred.shape[2]), min(W, pred.shape[3]), tgt, want_u8=True)
                        ss = fldr_hip.ssim_y_u8(img, tgt) if want_ssim else None
                        res.append((sse, ss))
                # one synchronisation per pair, after all its outputs are queued
                out = []
                for sse, ss in res:
                    mse = float(sse[0].item()) / (3.0 * H * W)
                    out.append((float("inf") if mse == 0 else 10 * math.log10(255.0 ** 2 / mse), float(ss[0].item()) if ss is not None else None))
                fldr_hip.check_range()
                return out
            finally:
                model.pair_cache = prev
                model._pair_state = None
    dev = device if device is not None else "cpu"
    sums = {"psnr": 0.0, "ssim": 0.0, "n": 0.0}
    per_t, scenes = {}, {}
    for i in shard_pairs(len(pairs), rank, world):
        p0, p1, scene, targets = pairs[i]
        u8 = torch.from_numpy(np.stack([load_bgr_u8(p0), load_bgr_u8(p1)], 0)).permute(0, 3, 1, 2).unsqueeze(0).contiguous().to(dev)
        tg = [torch.from_numpy(load_bgr_u8(pt)).permute(2, 0, 1).unsqueeze(0).contiguous().to(dev) for pt, _ in targets]
        for (pt, tv), (ps, ss) in zip(targets, predict(u8, [tv for _, tv in targets], tg)):
            sums["psnr"] += ps
            sums["ssim"] += ss if ss is not None else 0.0
            sums["n"] += 1
            per_t.setdefault(round(tv, 6), []).append(ps)
            scenes.setdefault(scene, []).append(ps)
            if log:
                log("%s %s t=%.3f PSNR %.3f%s" % (scene, os.path.basename(pt), tv, ps, "" if ss is None else " SSIM %.5f" % ss))
    tot = reduce_sums([sums["psnr"], sums["ssim"], sums["n"]], dev)
    keys_t = sorted({round(float(t), 6) for t in np.linspace(1 / multiple, 1 - 1 / multiple, multiple - 1)})
    pt = reduce_sums([sum(per_t.get(k, [])) for k in keys_t] + [len(per_t.get(k, [])) for k in keys_t], dev)
    n = max(tot[2], 1.0)
    return {"psnr": tot[0] / n, "ssim": (tot[1] / n) if want_ssim else None, "frames": int(tot[2]), "pairs": len(pairs),
            "per_t": {k: pt[j] / max(pt[len(keys_t) + j], 1.0) for j, k in enumerate(keys_t)},
            "scenes_this_rank": {k: sum(v) / len(v) for k, v in scenes.items()}}


# ---- data-parallel helpers (one process per GPU; frame pairs are independent: SURVEY 8e) -----------------
def shard_pairs(n_pairs, rank, world):
    """Static round-robin of frame PAIRS (not outputs) over ranks: pair i -> rank i % world, so that all t values
    of a pair stay on one rank."""
    return [i for i in range(n_pairs) if i % world == rank]


def _collectives_on():
    """True when a process group exists and its collectives should run: more than one rank, or FLDR_BENCH_FORCE_PG=1 (the
    world-size-1 RCCL rehearsal: the same all_reduce / all_gather calls on device tensors as on an 8-GPU node)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or os.environ.get("FLDR_BENCH_FORCE_PG") == "1"


def max_over_ranks(value, device="cpu"):
    """MAX-reduce a Python float over the default process group (no-op without one)."""
    import torch.distributed as dist
    if not _collectives_on():
        return float(value)
    t = torch.tensor([value], device=device, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t.item()


def reduce_psnr(psnr_sum, count, device="cpu"):
    """SUM-reduce (sum of PSNRs, number of frames) over ranks -> global mean PSNR: the only collective the
    evaluation loop needs (16 bytes; main.py:962 keeps these in AverageClass on a single process)."""
    import torch.distributed as dist
    t = torch.tensor([psnr_sum, float(count)], device=device, dtype=torch.float64)
    if _collectives_on():
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return (t[0] / t[1].clamp(min=1)).item(), int(t[1].item())


def gather_floats(value, device="cpu"):
    """One float per rank -> list over ranks (all_gather of 8 bytes; [value] without a process group)."""
    import torch.distributed as dist
    if not _collectives_on():
        return [float(value)]
    mine = torch.tensor([value], device=device, dtype=torch.float64)
    out = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    return [float(x.item()) for x in out]


GPU_BOX_CPU_SHARE = 16      # CPUs a one-GPU lease of the benchmark pool is entitled to when the host exposes all of its cores


def host_core_budget():
    """What this process may use of the host, stated rather than guessed: logical CPUs, the affinity mask, the cgroup
    CPU quota if one is set, and the thread count the CPU baseline runs with = min(affinity, quota).  Only when NO limit
    is discoverable on a large shared host (> 64 CPUs visible) the count is held to the pool's per-GPU share and
    `capped_to_share` says so."""
    n_all = os.cpu_count() or 1
    aff = n_all
    try:
        aff = len(os.sched_getaffinity(0))
    except Exception:
        pass
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    quota = max(1, int(int(txt[0]) / int(txt[1])))
            else:
                q = int(txt[0])
                if q > 0:
                    quota = max(1, q // int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read()))
        except Exception:
            pass
        if quota is not None:
            break
    used = min(aff, quota) if quota is not None else aff
    capped = False
    if quota is None and aff == n_all and used > 64:
        used, capped = GPU_BOX_CPU_SHARE, True
    return {"logical_cpus": n_all, "affinity": aff, "cgroup_quota": quota, "cores_used": used, "capped_to_share": capped}


def host_cores():
    """Threads the CPU baseline uses (see host_core_budget)."""
    return host_core_budget()["cores_used"]
