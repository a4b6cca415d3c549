"""The per-sequence driver: what finalposes.py:105-238 / choosePose.py:154-309 (per-image
registration), verfication.py:61-108 (consecutive-pair Chamfer pick) and icp.py:37-126 (ICP +
final Chamfer) do, as one asynchronous pipeline per image on the HIP stream and one reduction per
sequence.  Everything between the encoder output and the pose stays on the device: the reference's
idx.cpu() / nidx.cpu() round trips (inference.py:273-290) are gone because the kept count M and
the RANSAC status live in device memory.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
import torch

from . import ops, registration, shard


@dataclass
class SequenceModel:
    """First-sequence model as exported by genFeat.py:223-228 (vert1_scaled / feat1_scaled)."""
    keys: torch.Tensor   # (N, D) descriptors on the device, bf16 (MFMA bf16 path) or f32 (exact path)
    pts: torch.Tensor    # (N, 3) f32 surface points, mm
    log2_queries: bool = False   # bf16 queries carry a log2(e) prescale (ops.prescale_queries_log2)
    screened: bool = False       # with log2_queries: ISR_DTYPE_BF16_LOG2_SCREENED (K1 behind the FP6 screen; peaked softmaxes)


# True: the group routes (register_crops, register_block) let K1's epilogue form the first histogram of the top-80 % cut
# (isr_corr_argmax_digits -> isr_select_top_batch_digits): the same results, nine launches instead of ten and
# one read of logp less per group.  OFF by default: K1 is bound by its wave slots, and the counts — one device-scope integer
# atomic per wave and distinct digit, 770 000 per 32-image launch on ~20 addresses per image — hold the finishing waves' slots:
# 29.1 against 26.9 ms per launch alone, the step 5 % slower (profiles/r05_epilogue_histogram_ab.txt; bench.py --epilogue-digits).
EPILOGUE_DIGITS = False


@dataclass
class ImageResult:
    pose: torch.Tensor     # (3,4) f64 device [R|t]
    status: torch.Tensor   # (1,) i32 device
    n_inl: torch.Tensor    # (1,) i32 device
    inl_idx: torch.Tensor  # (P,) i32 device (first n_inl valid; indices into the kept set)
    keep: torch.Tensor     # (P,) i32 device (first M valid): nidx of inference.py:289
    M: torch.Tensor        # (1,) i32 device
    idx: torch.Tensor      # (P,) i32 device: idx1 of inference.py:273
    logp: torch.Tensor     # (P,) f32 device: in1[:,0]
    n_eval: torch.Tensor | None = None   # (1,) i32 device: hypotheses the staged RANSAC loop scored


def register_image(model: SequenceModel, queries: torch.Tensor, pix_xy: torch.Tensor, cam,
                   itr: int = 500, reperr: float = 2.0, seed: int = 0, refine_iters: int = 10,
                   timing: list | None = None, confidence: float = 0.99) -> ImageResult:
    """inference.py:273-293 for one image, fully enqueued (no host synchronisation):
    getCors -> top-80 % filter -> correspondence assembly -> pnp(itr, reperr, P3P)."""
    if timing is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    idx, logp = ops.corr_argmax(queries, model.keys, log2_prescaled=model.log2_queries, screened=model.screened and model.log2_queries)
    if timing is not None:
        e1.record()
        timing.append((e0, e1))
    keep, M, _ = ops.select_top(logp)
    p3d, p2d = ops.gather_corr(idx, keep, M, model.pts, pix_xy)
    r = ops.pnp_ransac(p3d, p2d, cam, H=itr, reperr=reperr, seed=seed, refine_iters=refine_iters, M_dev=M,
                       confidence=confidence)
    return ImageResult(r.pose, r.status, r.n_inl, r.inl_idx, keep, M, idx, logp, r.n_eval)


def register_crop(model: SequenceModel, feat: torch.Tensor, mask: torch.Tensor, cam, c0: int = 0,
                  n_feat: int | None = None, down_sample: int = 3, itr: int = 500, reperr: float = 2.0,
                  seed: int = 0, refine_iters: int = 10, confidence: float = 0.99) -> tuple[ImageResult, torch.Tensor]:
    """inference.py:248-293 from the network output on: `feat` = imfeatsfull (1, H, W, C) or (H, W, C)
    channels-last on the device, `mask` = cropMask (H, W[, 3]) uint8 on the device, `cam` the cropped
    and down-sampled camera matrix (formats.crop_camera).  Sub-sampling, masking, the compaction of the
    masked descriptors into K1's operand layout and the pixel list run on the device
    (isr_prep_queries); the number of masked pixels never visits the host.  Returns the ImageResult
    (arrays have the capacity ceil(H/ds) * ceil(W/ds); idx / logp rows past the count are padding) and
    the device count n_dev."""
    D = model.keys.shape[1] if n_feat is None else n_feat
    if model.keys.dtype == torch.bfloat16:
        dtype = "bf16_log2" if model.log2_queries else "bf16"
    else:
        dtype = "f32"
    Q, pix, n_dev = ops.prep_queries(feat, mask, c0=c0, D=D, step=down_sample, dtype=dtype)
    keys = model.keys if model.keys.shape[1] == Q.shape[1] else ops._pad_cols(model.keys, Q.shape[1])
    idx, logp = ops.corr_argmax(Q, keys, log2_prescaled=model.log2_queries, screened=model.screened and model.log2_queries)
    keep, M, _ = ops.select_top(logp, n_dev=n_dev)
    p3d, p2d = ops.gather_corr(idx, keep, M, model.pts, pix)
    r = ops.pnp_ransac(p3d, p2d, cam, H=itr, reperr=reperr, seed=seed, refine_iters=refine_iters, M_dev=M,
                       confidence=confidence)
    return ImageResult(r.pose, r.status, r.n_inl, r.inl_idx, keep, M, idx, logp, r.n_eval), n_dev


def register_frame(model: SequenceModel, rgb, mask, camparams, encoder, n_feat: int = 12, down_sample: int = 3,
                   itr: int = 500, reperr: float = 2.0, seed: int = 0, refine_iters: int = 10, confidence: float = 0.99,
                   useMask: bool = True):
    """One iteration of the reference's per-image loop, inference.py:196-293, from the raw frame to the pose with
    everything but the descriptor network in this package: the crop front end on the device
    (registration.crop_inputs: mask box, crop affine + camera matrix, warp of image and mask, useMask blanking,
    normalize), `encoder` — the caller's network, `encoder_rgb` of inference.py:237: (1, 3, 224, 224) f32 ->
    (1, C, 224, 224) with the descriptor in channels [0, n_feat) — and then register_crop (sub-sampling, masking,
    getCors, top-80 % filter, pnp).  Returns (ImageResult, n_dev, camMat (3, 3))."""
    from . import registration
    inputIM, cropMask, cam, _ = registration.crop_inputs(rgb, mask, camparams, useMask=useMask, down_sample=down_sample)
    with torch.no_grad():
        imfeatsfull = torch.movedim(encoder(inputIM), 1, 3)                       # inference.py:236-237
    res, n_dev = register_crop(model, imfeatsfull, cropMask[0], cam[0], n_feat=n_feat, down_sample=down_sample, itr=itr,
                               reperr=reperr, seed=seed, refine_iters=refine_iters, confidence=confidence)
    return res, n_dev, cam[0]


def register_crops(model: SequenceModel, feats: torch.Tensor, masks: torch.Tensor, cams, c0: int = 0,
                   n_feat: int | None = None, down_sample: int = 3, itr: int = 500, reperr: float = 2.0,
                   seeds=None, refine_iters: int = 10, confidence: float = 0.99, group: int = 128,
                   n_streams: int = 1) -> tuple[list[ImageResult], torch.Tensor]:
    """The reference's per-image loop (inference.py:163, 248-293) from the network output on, BATCHED: `feats`
    (n, H, W, C) channels-last on the device, `masks` (n, H, W[, 3]) uint8 on the device, `cams` (n, 3, 3) the
    cropped, down-sampled camera matrices (formats.crop_camera).  Per `group` images: ONE isr_prep_queries_batch
    (three launches), ONE K1 launch on group * S capacity rows (S = ceil(H/ds) * ceil(W/ds); rows past an image's
    count are zero queries whose results nobody reads), ONE filter / assembly / RANSAC chain (register_group with
    the ragged counts).  At the reference's shape (75 x 75 crop, D = 12, N = 80 000) a single image is bound by
    ~45 dependent launches of a few microseconds of work each; a group shares them (0.6 launches per image at
    group = 64; 128-256 crops per group fill the chip best: K1's ~9 working workgroups per crop against 768 slots).  n_streams > 1 runs K1 of group g+1 on its own stream beside the chain of group g (slower at this
    shape, see below).  Every image's outputs are bit-identical to register_crop's.
    Returns (results, n_dev (n,) i32 on the device)."""
    dev = model.keys.device
    n = feats.shape[0]
    D = model.keys.shape[1] if n_feat is None else n_feat
    if model.keys.dtype == torch.bfloat16:
        dtype = "bf16_log2" if model.log2_queries else "bf16"
    else:
        dtype = "f32"
    cams = np.asarray(cams, np.float64)
    cams = np.broadcast_to(cams, (n, 3, 3)) if cams.ndim == 2 else cams
    seeds = list(range(n)) if seeds is None else list(seeds)
    cur = torch.cuda.current_stream(dev)
    # n_streams <= 1 (the default here): everything on the caller's stream.  At this shape a K1 workgroup lives for
    # milliseconds (256 queries x 80 000 keys) and fills its CU's registers; a chain of ~50 small DEPENDENT launches
    # on a side stream then waits for a K1 workgroup to retire before each of its kernels can start (measured:
    # 207 us per gn_accumulate launch instead of ~8) and the chain, not K1, bounds the group.  In sequence the chain
    # costs ~0.4 ms per group behind a ~3 ms K1.  (bench.py's 640 x 480 shape is the opposite case: K1 is 30 ms
    # per launch and the chain hides beside it — sequence.register_block keeps its side streams.)
    serial = n_streams <= 1
    pool = [cur, cur] if serial else _stream_pool(dev, max(n_streams, 2))
    if not serial:
        for s in pool:
            s.wait_stream(cur)
    k1_stream, side = pool[0], pool[1:]
    out, counts, keys = [], [], None
    for gi, g0 in enumerate(range(0, n, group)):
        g1 = min(n, g0 + group)
        B = g1 - g0
        with torch.cuda.stream(k1_stream):
            Q, pix, n_dev = ops.prep_queries_batch(feats[g0:g1], masks[g0:g1], c0=c0, D=D, step=down_sample, dtype=dtype)
            if keys is None:
                keys = model.keys if model.keys.shape[1] == Q.shape[2] else ops._pad_cols(model.keys, Q.shape[2])
            S = Q.shape[1]
            # (the cut's first histogram is formed in K1's epilogue, of each crop's n_dev live rows: SURVEY 8(f)-2)
            if EPILOGUE_DIGITS:
                idx_g, logp_g, hist_g = ops.corr_argmax(Q.view(B * S, -1), keys, log2_prescaled=model.log2_queries,
                                                        screened=model.screened and model.log2_queries, rows_per_image=S, n_rows=n_dev)
            else:
                idx_g, logp_g = ops.corr_argmax(Q.view(B * S, -1), keys, log2_prescaled=model.log2_queries,
                                                screened=model.screened and model.log2_queries)
                hist_g = None
            if not serial:
                done = torch.cuda.Event()
                done.record(k1_stream)
        s = side[gi % len(side)]
        if not serial:
            s.wait_event(done)
            for t in (idx_g, logp_g, hist_g, pix, n_dev):
                if t is not None:
                    t.record_stream(s)
        with torch.cuda.stream(s):
            keep, M, _ = ops.select_top_batch(logp_g.view(B, S), n_dev=n_dev, digit_hist=hist_g)
            p3d, p2d = ops.gather_corr_batch(idx_g.view(B, S), keep, M, model.pts, pix)
            r = ops.pnp_ransac_batch(p3d, p2d, cams[g0:g1], M, H=itr, reperr=reperr, seeds=seeds[g0:g1],
                                     refine_iters=refine_iters, confidence=confidence)
        iv, lv = idx_g.view(B, S), logp_g.view(B, S)
        out += [ImageResult(r.pose[b], r.status[b:b + 1], r.n_inl[b:b + 1], r.inl_idx[b], keep[b], M[b:b + 1], iv[b], lv[b],
                            r.n_eval[b:b + 1]) for b in range(B)]
        counts.append(n_dev)
    if not serial:
        for s in pool:
            cur.wait_stream(s)
        _publish(out, cur)
        for n_dev in counts:              # allocated on k1_stream, read by the cat below on the caller's stream
            n_dev.record_stream(cur)
    n_all = torch.cat(counts)
    return out, n_all


def register_frames(model: SequenceModel, rgbs, masks, camparams, encoder, n_feat: int = 12, down_sample: int = 3,
                    itr: int = 500, reperr: float = 2.0, seeds=None, refine_iters: int = 10, confidence: float = 0.99,
                    useMask: bool = True, group: int = 128):
    """The reference's per-image loop, inference.py:163-293, for a block of frames: the crop front end of every
    frame on the device in two launches (registration.crop_inputs: mask boxes, crop affines + camera matrices,
    warps, useMask blanking, normalize), ONE call of `encoder` on the (n, 3, 224, 224) batch (the caller's network,
    `encoder_rgb` of inference.py:237), then register_crops.  Returns (results, n_dev (n,), camMat (n, 3, 3))."""
    inputIM, cropMask, cam, _ = registration.crop_inputs(rgbs, masks, camparams, useMask=useMask, down_sample=down_sample)
    with torch.no_grad():
        imfeatsfull = torch.movedim(encoder(inputIM), 1, 3)                       # inference.py:236-237
    res, n_dev = register_crops(model, imfeatsfull, cropMask, cam, n_feat=n_feat, down_sample=down_sample, itr=itr,
                                reperr=reperr, seeds=seeds, refine_iters=refine_iters, confidence=confidence, group=group)
    return res, n_dev, cam


def _publish(results: list[ImageResult], consumer: torch.cuda.Stream) -> None:
    """The tensors of `results` were allocated on side streams and are about to be read on `consumer`
    (which has been made to wait for those streams).  Tell the caching allocator: without
    record_stream a dropped result's block goes back to its side stream's pool at once, and the next
    step's allocations there may overwrite it while `consumer` is still reading."""
    for r in results:
        for t in (r.pose, r.status, r.n_inl, r.inl_idx, r.keep, r.M, r.idx, r.logp, r.n_eval):
            if t is not None:
                t.record_stream(consumer)


_streams: dict[tuple, list] = {}
_closed: dict[tuple, torch.cuda.Event] = {}       # (device, workspace tag) -> the event behind the last close on that workspace

# register_block: a group's K1 call is OPENED on the K1 stream (key norms + the chip-filling kernel) and CLOSED on the group's
# side stream (fallback, finalize, recheck, merge: ~170 us of small launches at low occupancy, ahead of the cut that reads their
# outputs), so that the next group's chip-filling kernel follows this one's directly: on one stream the closing kernels and the
# next call's key-norm kernel stood between them (K1's duty cycle 0.977, profiles/r05_side_work_budget.txt).  Two workspaces
# alternate; an opening waits for the close that last used its workspace.  The chip-filling kernels stay one behind the other on
# one stream (their durations mean what they say).  False: one isr_corr_argmax call per group.  Same results either way.
K1_SPLIT_CLOSE = True


def _stream_pool(dev: torch.device, n: int) -> list:
    key = (dev.index, n)
    if key not in _streams:
        # stream 0 carries K1; the side streams carry the per-image chains of small dependent launches
        # and get the higher priority: they need a few CUs for microseconds, and every microsecond
        # they wait for a slot is serial latency (measured: +2 % images/s, K1-high: -4 %)
        _streams[key] = [torch.cuda.Stream(device=dev, priority=0 if i == 0 else -1) for i in range(n)]
    return _streams[key]


def register_images(model: SequenceModel, images, cam, itr: int = 500, reperr: float = 2.0,
                    seed0: int = 0, refine_iters: int = 10, n_streams: int = 3,
                    confidence: float = 0.99) -> list[ImageResult]:
    """Register a block of images with the stages pipelined over HIP streams.

    Stream 0 runs nothing but K1 (getCors) back to back — the only kernel that fills the chip;
    image j's filter + assembly + RANSAC chain (a dozen small dependent launches) waits on K1(j)'s
    event and runs on side stream 1 + j % (n_streams - 1), beside K1(j+1) and inside K1's launch
    tail (a 640x480 image is 1.17 rounds of resident workgroups).  Events recorded around K1 on
    stream 0 therefore bracket the kernel alone.  `images` is a sequence of (queries, pix_xy);
    `cam` one 3x3 or one per image.  The caller's current stream waits for every stream before
    this returns; there is no host synchronisation."""
    dev = model.keys.device
    cur = torch.cuda.current_stream(dev)
    cam_of = (lambda j: cam) if np.ndim(cam) == 2 else (lambda j: cam[j])
    if n_streams <= 1:
        return [register_image(model, q, pix, cam_of(j), itr, reperr, seed0 + j, refine_iters, confidence=confidence)
                for j, (q, pix) in enumerate(images)]
    pool = _stream_pool(dev, n_streams)
    for s in pool:
        s.wait_stream(cur)
    k1_stream, side = pool[0], pool[1:]
    out = []
    for j, (q, pix) in enumerate(images):
        with torch.cuda.stream(k1_stream):
            idx, logp = ops.corr_argmax(q, model.keys, log2_prescaled=model.log2_queries, screened=model.screened and model.log2_queries)
            done = torch.cuda.Event()
            done.record(k1_stream)
        s = side[j % len(side)]
        s.wait_event(done)
        idx.record_stream(s)
        logp.record_stream(s)
        with torch.cuda.stream(s):
            keep, M, _ = ops.select_top(logp)
            p3d, p2d = ops.gather_corr(idx, keep, M, model.pts, pix)
            r = ops.pnp_ransac(p3d, p2d, cam_of(j), H=itr, reperr=reperr, seed=seed0 + j,
                               refine_iters=refine_iters, M_dev=M, confidence=confidence)
        out.append(ImageResult(r.pose, r.status, r.n_inl, r.inl_idx, keep, M, idx, logp, r.n_eval))
    for s in pool:
        cur.wait_stream(s)
    _publish(out, cur)
    return out


def register_group(model: SequenceModel, idx_g: torch.Tensor, logp_g: torch.Tensor, pix_xy: torch.Tensor, cams,
                   itr: int, reperr: float, seeds, refine_iters: int, confidence: float = 0.99,
                   digit_hist: torch.Tensor | None = None) -> list[ImageResult]:
    """inference.py:282-293 for a GROUP of images whose K1 results are idx_g / logp_g (B, P): the
    top-80 % filter, the correspondence assembly and pnp() run as ONE chain of launches with the
    image on blockIdx.z (isr_select_top_batch, isr_gather_corr_batch, isr_pnp_ransac_batch) — the
    same kernels as the per-image calls, so every image's outputs are bit-identical to
    register_image's.  pix_xy (P, 2) shared or (B, P, 2); cams one 3x3 or (B, 3, 3).  digit_hist (B, 2048): the cut's first
    histogram when K1 formed it (ops.corr_argmax(..., rows_per_image=P))."""
    B = idx_g.shape[0]
    keep, M, _ = ops.select_top_batch(logp_g, digit_hist=digit_hist)
    p3d, p2d = ops.gather_corr_batch(idx_g, keep, M, model.pts, pix_xy)
    r = ops.pnp_ransac_batch(p3d, p2d, cams, M, H=itr, reperr=reperr, seeds=seeds, refine_iters=refine_iters,
                             confidence=confidence)
    return [ImageResult(r.pose[b], r.status[b:b + 1], r.n_inl[b:b + 1], r.inl_idx[b], keep[b], M[b:b + 1], idx_g[b], logp_g[b],
                        r.n_eval[b:b + 1])
            for b in range(B)]


def register_block(model: SequenceModel, queries: torch.Tensor, pix_xy: torch.Tensor, cam, itr: int = 500,
                   reperr: float = 2.0, seed0: int = 0, refine_iters: int = 10, n_streams: int = 3,
                   group: int = 8, confidence: float = 0.99) -> list[ImageResult]:
    """register_images for a block held as ONE tensor: queries (n, P, D), pix_xy (n, P, 2) or (P, 2).
    K1 runs once per `group` images on (group * P) query rows — K1's result for a query does not
    depend on the launch it rides in, so this only changes the launch shape: a 640x480 image alone is
    1.17 rounds of resident workgroups, sixteen together are 18.8, and the launch tail shrinks from
    ~20 % to ~2 % of K1's time.  The group's filter / assembly / RANSAC chain (register_group: ~30
    launches for the whole group instead of ~35 per image) runs on a side stream beside the next K1
    launch; groups alternate between the side streams."""
    dev = model.keys.device
    n, P = queries.shape[0], queries.shape[1]
    cur = torch.cuda.current_stream(dev)
    cams = np.asarray(cam, np.float64)
    cams = np.broadcast_to(cams, (n, 3, 3)) if cams.ndim == 2 else cams
    pool = _stream_pool(dev, max(n_streams, 2))
    for s in pool:
        s.wait_stream(cur)
    k1_stream, side = pool[0], pool[1:]
    kw = dict(log2_prescaled=model.log2_queries, screened=model.screened and model.log2_queries)
    if EPILOGUE_DIGITS:
        kw["rows_per_image"] = P
    out = []
    for gi, g0 in enumerate(range(0, n, group)):
        g1 = min(n, g0 + group)
        rows = queries[g0:g1].reshape((g1 - g0) * P, -1)
        with torch.cuda.stream(k1_stream):
            if K1_SPLIT_CLOSE:
                tag = f"corr{gi & 1}"
                prev = _closed.get((dev.index, tag))
                if prev is not None:
                    k1_stream.wait_event(prev)             # the workspace's last close (two groups ago: long finished)
                call = ops.corr_argmax_open(rows, model.keys, ws_tag=tag, **kw)
                res = call.outputs()
            else:
                res = ops.corr_argmax(rows, model.keys, **kw)
            done = torch.cuda.Event()
            done.record(k1_stream)
        s = side[gi % len(side)]
        s.wait_event(done)
        for t in res:
            t.record_stream(s)
        with torch.cuda.stream(s):
            if K1_SPLIT_CLOSE:
                ops.corr_argmax_close(call)
                closed = torch.cuda.Event()
                closed.record(s)
                _closed[(dev.index, tag)] = closed
            idx_g, logp_g = res[0], res[1]
            out += register_group(model, idx_g.view(g1 - g0, P), logp_g.view(g1 - g0, P),
                                  pix_xy if pix_xy.ndim == 2 else pix_xy[g0:g1], cams[g0:g1], itr, reperr,
                                  [seed0 + j for j in range(g0, g1)], refine_iters, confidence,
                                  digit_hist=res[2] if EPILOGUE_DIGITS else None)
    for s in pool:
        cur.wait_stream(s)
    _publish(out, cur)
    return out


def stack_poses(results: list[ImageResult]) -> tuple[torch.Tensor, torch.Tensor]:
    """(n,12) f64 poses and (n,) status, still on the device."""
    return (torch.stack([r.pose.reshape(12) for r in results]),
            torch.cat([r.status for r in results]))


def acceptance_counts(model_verts, surface_pts, R_gt, t_gt, poses: torch.Tensor, status: torch.Tensor, diameter: float,
                      dataset: str = "tless", names=None) -> dict:
    """The per-image acceptance bookkeeping of inference.py:300-320 for a whole block of registered images in two
    launches instead of two KD-tree builds per image:
        final_error  = ADDS(modelVerts, gtR, gtT, R2, T2)            (dataset == "tless";  ADD otherwise)
        final_errorR = ADDS(modelVerts, gtR, 0, R2, 0)               (rotation only)
        workCT += final_error < 0.1 * diameter;  rotWorkCT += final_errorR < 0.1 * diameter
        correct_predicted_ids.append(name)  for accepted images      (-> correctly_predicted_list.txt, :369-374)
    poses (n, 12) [R|t] f64 and status (n,) on the device (sequence.stack_poses).  An image whose pnp failed (status
    0; the reference's `(1, 1, 1)` sentinel would raise inside ADDS there) is never accepted: its errors are inf.
    Returns {final_error (n,), final_errorR (n,), work (n,) bool, rot_work (n,) bool, workCT, rotWorkCT,
    correct_predicted_ids}."""
    v = registration._dev(model_verts, torch.float32)
    n = poses.shape[0]
    Tp = poses.reshape(n, 3, 4).to(torch.float64)
    Tg = torch.cat([registration._dev(R_gt, torch.float64).reshape(n, 3, 3),
                    registration._dev(t_gt, torch.float64).reshape(n, 3, 1)], dim=2)
    Tp0, Tg0 = Tp.clone(), Tg.clone()
    Tp0[:, :, 3] = 0.0
    Tg0[:, :, 3] = 0.0
    if dataset == "tless":
        sp = registration._dev(surface_pts, torch.float32)
        # queries = GT-pose CAD vertices, targets = predicted-pose surface points (inference.py:118-120)
        err = ops.nn_batched(v, sp, Tg.reshape(n, 12), Tp.reshape(n, 12)).sum_d / v.shape[0]
        errR = ops.nn_batched(v, sp, Tg0.reshape(n, 12), Tp0.reshape(n, 12)).sum_d / v.shape[0]
    else:
        err = ops.add_metric(v, Tg.reshape(n, 12), Tp.reshape(n, 12))
        errR = ops.add_metric(v, Tg0.reshape(n, 12), Tp0.reshape(n, 12))
    ok = status.reshape(n) != 0
    inf = torch.full_like(err, float("inf"))
    err, errR = torch.where(ok, err, inf), torch.where(ok, errR, inf)
    host = torch.stack([err, errR]).cpu().numpy()           # one copy
    work, rot = host[0] < 0.1 * diameter, host[1] < 0.1 * diameter
    ids = [names[i] if names is not None else i for i in np.nonzero(work)[0]]
    return dict(final_error=host[0], final_errorR=host[1], work=work, rot_work=rot, workCT=int(work.sum()),
                rotWorkCT=int(rot.sum()), correct_predicted_ids=ids)


def _orthonormal(R, tol=1e-9) -> bool:
    """R2 inv(R1) = R2 R1^T exactly when R1 is orthonormal (then np.linalg.inv and the transpose agree
    to rounding); scene_gt rotations with few printed digits fall back to the general inverse."""
    return bool(np.all(np.abs(np.einsum("nij,nkj->nik", R, R) - np.eye(3)) < tol))


def pick_by_chamfer(pc1: torch.Tensor, poses_all: torch.Tensor, R_gt_all: np.ndarray, t_gt_all: np.ndarray,
                    n_total: int) -> tuple[int, float]:
    """verfication.py:61-108 sharded: this rank evaluates the consecutive pairs it owns, then ONE
    all-reduce(MIN) over the (n - 1)-entry f64 table (shard.allreduce_min_table) hands every rank the whole
    `chamferdis` list, and min / list.index(min) — the first minimum, in f64 — is taken on every rank
    identically.  poses_all (n,12) predicted poses of ALL images (after the all-gather), R_gt_all/t_gt_all
    the GT poses from scene_gt.json.  Returns (index, value)."""
    idx, val, _ = pick_by_chamfer_table(pc1, poses_all, R_gt_all, t_gt_all, n_total)
    return idx, val


def chamfer_pairs_owned(pc1: torch.Tensor, poses_all: torch.Tensor, R_gt_all: np.ndarray, t_gt_all: np.ndarray,
                        lo: int, hi: int) -> torch.Tensor:
    """The device part of the sharded pick: the Chamfer values of the consecutive pairs (i, i+1), i in [lo, hi)
    (verfication.py:70-102) as a (hi - lo,) f64 device tensor.  A pair's value is a function of the two images'
    poses only, whichever block it is evaluated in."""
    if hi <= lo:
        return torch.empty(0, dtype=torch.float64, device=pc1.device)
    Rp = poses_all.reshape(-1, 3, 4)[lo:hi + 1, :, :3].cpu().numpy()
    # rotation block of [R2|T2] inv([R1|T1]) (verfication.py:9-19) for every owned pair at once
    Rg = np.asarray(R_gt_all, np.float64)
    Rrel = np.einsum("nij,nkj->nik", Rg[lo + 1:hi + 1], Rg[lo:hi]) if _orthonormal(Rg[lo:hi]) else np.stack(
        [registration.calculate_relative_pose(R_gt_all[i], t_gt_all[i], R_gt_all[i + 1], t_gt_all[i + 1])[0]
         for i in range(lo, hi)])
    return registration.chamfer_pairs(pc1, Rp, Rrel)


def pick_by_chamfer_table(pc1: torch.Tensor, poses_all: torch.Tensor, R_gt_all: np.ndarray, t_gt_all: np.ndarray,
                          n_total: int) -> tuple[int, float, np.ndarray]:
    """pick_by_chamfer, also returning the full chamferdis table (n - 1,) f64 as a NumPy array."""
    rank, size = shard.world()
    lo, hi = shard.owned_pairs(n_total, rank, size)
    ch = chamfer_pairs_owned(pc1, poses_all, R_gt_all, t_gt_all, lo, hi)
    table = shard.allreduce_min_table(ch, lo, n_total - 1).cpu().numpy()
    idx, val = shard.first_min(table)
    return idx, val, table


_field_cache: dict = {}


def cloud_key(pts: torch.Tensor) -> tuple:
    """(hash1, hash2, shape, dtype, device index) of a cloud's bytes: see surface_field."""
    flat = pts.detach().contiguous().view(-1)
    bits = flat.view(torch.int32 if flat.element_size() == 4 else torch.int64).to(torch.int64)
    i = torch.arange(bits.numel(), device=bits.device, dtype=torch.int64)
    h1 = (bits * (2 * i + 1)).sum()                                    # int64 arithmetic wraps: sums modulo 2^64
    h2 = ((bits ^ (bits >> 15)) * (i * 0x9E3779B1 + 0x7F4A7C15 | 1)).sum()
    h = torch.stack([h1, h2]).cpu().numpy()
    return (int(h[0]), int(h[1]), tuple(pts.shape), str(pts.dtype), pts.device.index)


def surface_field(surface_pts: torch.Tensor, cells: int = 128) -> "ops.DistField":
    """The distance field of a surface cloud for vote_rows' bounds, built once per cloud."""
    # keyed by CONTENT (one small read-back): an address can be handed to another cloud of the same shape.  The key is a pair of
    # position-weighted wrapping int64 sums over the coordinates' BIT patterns (element i weighted by an odd multiplier that
    # depends on i), so a cloud with permuted axes, swapped rows or one changed coordinate gets another key (round 4 keyed on
    # two f64 sums that a within-row permutation left unchanged).
    key = cloud_key(surface_pts) + (cells,)
    fld = _field_cache.get(key)
    if fld is None:
        if len(_field_cache) > 4:
            _field_cache.clear()
        fld = _field_cache[key] = ops.dist_field(surface_pts, cells=cells)
    return fld


VOTE_SLACK_MM = 1e-3      # an item within this of the threshold is always evaluated exactly (f32 coordinates, field rounding)


def vote_rows(model_verts, surface_pts, R_gt, t_gt, R_pred, t_pred, diameter, lo: int, hi: int, chunk: int = 4096,
              bounds: bool | None = None, stats: dict | None = None):
    """The device part of the row-sharded vote (choosePose.py:98-107, 121-138): rows [lo, hi) of
        error[i][j] = ADDS(modelVerts, gt_rel[i][j], pred_rel[i][j]) < 0.1 * diameter
    Returns (error rows (hi - lo, n) bool device tensor, their int32 row sums (hi - lo, 1)).
    The vote needs the DECISION, not the ADD-S value.  bounds (default: on from 1 024 items): every item first gets a
    rigorous lower and upper bound of its ADD-S from a distance field of the surface cloud (ops.adds_bounds: one gather per
    vertex); only items whose bounds straddle 0.1 * diameter (+- VOTE_SLACK_MM) go through the nearest-neighbour search —
    the booleans are those of evaluating every item (tested).  stats (optional dict) receives the counts."""
    n = len(R_gt)
    v = registration._dev(model_verts, torch.float32)
    sp = registration._dev(surface_pts, torch.float32)
    Rg, tg = registration._dev(R_gt, torch.float64), registration._dev(t_gt, torch.float64)     # arrays or tensors
    Rp, tp = registration._dev(R_pred, torch.float64), registration._dev(t_pred, torch.float64)
    gt_rel = ops.rel_pose_table(Rg, tg, 0, lo, hi).reshape(-1, 12)        # (rows * n, 12) f64, [R_i^T R_j | t_j - t_i]
    pr_rel = ops.rel_pose_table(Rp, tp, 0, lo, hi).reshape(-1, 12)
    items = (hi - lo) * n
    thr = 0.1 * float(diameter)
    V = v.shape[0]

    def exact(sel_g, sel_p):
        parts = [ops.nn_batched(v, sp, sel_g[s0:s0 + chunk], sel_p[s0:s0 + chunk]).sum_d for s0 in range(0, sel_g.shape[0], chunk)]
        return torch.cat(parts) / V

    use_bounds = items >= 1024 if bounds is None else bool(bounds)
    if not use_bounds:
        err_d = (exact(gt_rel, pr_rel) < thr).reshape(hi - lo, n)
        if stats is not None:
            stats.update(items=items, by_bounds=0, exact=items)
        return err_d, err_d.sum(dim=1, dtype=torch.int32)[:, None]
    fld = surface_field(sp)
    lb, ub = ops.adds_bounds(v, gt_rel, pr_rel, fld)
    accept = ub / V < thr - VOTE_SLACK_MM
    reject = lb / V > thr + VOTE_SLACK_MM
    open_idx = torch.nonzero(~(accept | reject))[:, 0]                    # the one host round trip: how many need the search
    err_flat = accept.clone()
    if open_idx.numel():
        err_flat[open_idx] = exact(gt_rel[open_idx], pr_rel[open_idx]) < thr
    if stats is not None:
        stats.update(items=items, by_bounds=items - int(open_idx.numel()), exact=int(open_idx.numel()),
                     accepted_by_bound=int(accept.sum().item()), rejected_by_bound=int(reject.sum().item()))
    err_d = err_flat.reshape(hi - lo, n)
    return err_d, err_d.sum(dim=1, dtype=torch.int32)[:, None]


def vote_choose_image(model_verts, surface_pts, R_gt, t_gt, R_pred, t_pred, diameter, top: int = 50, chunk: int = 4096,
                      bounds: bool | None = None, stats: dict | None = None):
    """choosePose.py:79-151 sharded by rows: every rank holds all n predicted and GT
    poses, builds rows block_range(n, rank, world) of the two relative-pose tables on the device
    (compute_rel_poses, choosePose.py:43-51), evaluates its rows of
        error[i][j] = ADDS(modelVerts, gt_rel[i][j], pred_rel[i][j]) < 0.1 * diameter
    with the batched NN kernel, and all-gathers the int32 row sums; argmax / top-50 are then computed
    identically on every rank (ties -> lower index).  The tables, the ADD-S values and the comparison stay on
    the device: one copy brings back this rank's error rows (the reference writes them to error.npy), one the
    row sums.  bounds / stats: see vote_rows (items decided from the distance field, the search only where needed).  Returns (image_id, top indices, local error rows (rows, n) f64 of 0/1)."""
    rank, size = shard.world()
    n = len(R_gt)
    lo, hi = shard.block_range(n, rank, size)
    dev = registration.device()
    if hi > lo:
        err_d, sums_local = vote_rows(model_verts, surface_pts, R_gt, t_gt, R_pred, t_pred, diameter, lo, hi, chunk, bounds, stats)
        err = err_d.to(torch.float64).cpu().numpy()
    else:
        err = np.zeros((0, n))
        sums_local = torch.zeros((0, 1), dtype=torch.int32, device=dev)
    if size > 1 and shard._coll_device().type == "cpu":
        sums_local = sums_local.cpu()
    sums = shard.allgather_rows(sums_local, n)[:, 0].cpu().numpy().astype(np.float64)
    image_id = int(np.argmax(sums))
    return image_id, np.argsort(-sums, kind="stable")[:top], err
