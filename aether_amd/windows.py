"""Sliding-window driver for long-video reconstruction and its multi-GPU sharding.

Reference behaviour (/root/reference/scripts/demo.py:235-251, 607-631): a video longer than `num_frames` is cut into
41-frame windows starting every `sliding_window_stride` frames (plus one tail window flush with the end); every window
is a full, independent pipeline call with a FRESH generator seeded with the same seed (D:629); the windows are then
blended on the CPU, sequentially (D:254-422).  The reference runs the windows one after the other on one GPU.

Here windows are independent units: with one process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI on
MI355X, "gloo" in the CPU tests) window w runs on rank w mod N, weights are replicated and nothing is exchanged during
denoising.  The only collective is ONE gather to rank 0 of the finished per-window outputs (rgb 170 MB + disparity 57 MB +
raymap 5 MB fp32 per window, packed on the device) so that rank 0 can run the sequential blend exactly as the reference does.  Results are
bit-identical to the single-process run for any N (same seeds, same per-window arithmetic, gather only moves data).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Callable, List, Optional, Sequence

import numpy as np
import torch


def get_window_starts(total_frames: int, sliding_window_size: int, temporal_stride: int) -> List[int]:
    """D:235-251: regular starts every `temporal_stride`, plus a last window flush with the end of the video."""
    last = total_frames - sliding_window_size
    starts = list(range(0, last + 1, temporal_stride))
    if total_frames > sliding_window_size and last % temporal_stride != 0:
        starts.append(last)
    return starts


@dataclass
class WindowResult:
    """One window's outputs.  Contract: `rgb` / `disparity` are numpy float32 arrays, EXCEPT after `run_windows(keep_on_device=True)`
    on a CUDA gather device, where they are float32 torch tensors on that device (views of the gathered buffers, for
    `blend_and_merge_window_results(..., device=)`); `raymap` is numpy (only `run_windows_merged` hands `WindowMerger.add` a device tensor, which
    the merger fetches asynchronously)."""
    start: int
    rgb: np.ndarray         # [F, H, W, 3] float32 (a torch tensor after run_windows(keep_on_device=True))
    disparity: np.ndarray   # [F, H, W]    float32 (likewise)
    raymap: np.ndarray      # [F, 6, h, w] float32


def _dist():
    import torch.distributed as dist
    return dist if (dist.is_available() and dist.is_initialized()) else None


def shard(items: Sequence, rank: int, world: int) -> List:
    """Round-robin assignment: item i -> rank i mod world (8 windows on 8/4/2/1 GPUs = 1/2/4/8 windows per rank)."""
    return [x for i, x in enumerate(items) if i % world == rank]


def _as_tensor(a) -> torch.Tensor:
    return a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


def run_windows(call_window: Callable[[int], "object"], starts: Sequence[int], gather_device: Optional[torch.device] = None,
                keep_on_device: bool = False, force_collective: bool = False) -> Optional[List[WindowResult]]:
    """Runs `call_window(start)` (one pipeline call returning .rgb/.disparity/.raymap as numpy arrays or torch tensors) for this
    rank's share of `starts`, then GATHERS the outputs on rank 0 (the only rank that merges).  Returns the complete, start-ordered
    list on rank 0 (None on other ranks).  Without an initialised process group it simply runs every window in order.

    The exchange is one `dist.gather(dst=0)` of a [windows-per-rank, 232 MB] fp32 payload per rank: under "nccl" (= RCCL, xGMI)
    the payload is packed on the GPU from the pipeline's device-resident outputs (`pipeline.keep_outputs_on_device`) — no numpy
    round trip — and only rank 0 receives (7 peer transfers of 232 MB per window instead of an all-gather to all 8 ranks); under
    "gloo" (CPU tests) it is a CPU tensor.  `gather_device` defaults to the current CUDA device under nccl.
    `keep_on_device`: rgb / disparity of the results stay torch tensors on the gather device (views of the gathered buffers) for
    `blend_and_merge_window_results(..., device=)`; the raymaps (pose algebra runs on the host) are numpy either way.
    `force_collective` runs the gather even in a one-rank group (lets a 1-GPU box exercise the RCCL path)."""
    dist = _dist()
    rank, world = (dist.get_rank(), dist.get_world_size()) if dist is not None else (0, 1)
    mine = shard(list(enumerate(starts)), rank, world)
    local = []
    for idx, s in mine:
        out = call_window(s)
        local.append((idx, s, _as_tensor(out.rgb), _as_tensor(out.disparity), _as_tensor(out.raymap)))

    def result(s, rgb, disp, ray, dev_views):
        # tensors are kept only when they really live on a GPU: under gloo / with CPU inputs the fields are numpy like the
        # reference's, so host-only consumers (the numpy blend) never meet a torch tensor
        if keep_on_device and dev_views and rgb.is_cuda and disp.is_cuda:
            return WindowResult(s, rgb, disp, ray.cpu().numpy().copy())
        return WindowResult(s, rgb.cpu().numpy(), disp.cpu().numpy(), ray.cpu().numpy())

    if dist is None or (world == 1 and not force_collective):
        return [result(s, a, b, c, a.is_cuda) for _, s, a, b, c in local]

    # ---- one gather to rank 0: every rank contributes ceil(n/world) slots (unused slots: index -1) ---------------------
    per_rank = (len(starts) + world - 1) // world
    shapes = (tuple(local[0][2].shape), tuple(local[0][3].shape), tuple(local[0][4].shape)) if local else None
    all_shapes = [None] * world
    dist.all_gather_object(all_shapes, shapes)
    shapes = next(s for s in all_shapes if s is not None)
    sizes = [int(np.prod(s)) for s in shapes]
    if dist.get_backend() == "nccl":
        dev = torch.device(gather_device) if gather_device is not None else torch.device("cuda", torch.cuda.current_device())
    else:
        dev = torch.device("cpu")
    payload = torch.empty(per_rank, sum(sizes), dtype=torch.float32, device=dev)
    index = torch.full((per_rank,), -1, dtype=torch.int64, device=dev)
    for slot, (idx, s, a, b, c) in enumerate(local):
        torch.cat([a.reshape(-1).to(dev, torch.float32), b.reshape(-1).to(dev, torch.float32), c.reshape(-1).to(dev, torch.float32)],
                  out=payload[slot])
        index[slot] = idx
    if len(local) < per_rank:
        payload[len(local):].zero_()
    gathered = [torch.empty_like(payload) for _ in range(world)] if rank == 0 else None
    gathered_idx = [torch.empty_like(index) for _ in range(world)] if rank == 0 else None
    dist.gather(payload, gathered, dst=0)
    dist.gather(index, gathered_idx, dst=0)
    if rank != 0:
        return None
    results: List[Optional[WindowResult]] = [None] * len(starts)
    for g, gi in zip(gathered, gathered_idx):
        for slot, idx in enumerate(gi.cpu().tolist()):
            if idx < 0:
                continue
            a, b, c = torch.split(g[slot], sizes)
            results[idx] = result(starts[idx], a.view(shapes[0]), b.view(shapes[1]), c.view(shapes[2]), True)
    assert all(r is not None for r in results)
    return results


def run_windows_merged(call_window: Callable[[int], "object"], starts: Sequence[int], *, height: int, width: int,
                       gather_device: Optional[torch.device] = None, smooth_camera: bool = True, smooth_method: str = "kalman",
                       out_dtype=np.float64, pinned: bool = False, force_collective: bool = False, timings: Optional[dict] = None):
    """Windows + exchange + merge as ONE pipeline (what scripts/demo.py and bench.py run for a long clip): the windows are processed in ROUNDS of
    one window per rank (window w on rank w mod N, as in `run_windows`); after each round ONE `dist.gather(dst=0)` moves that round's outputs
    (232 MB fp32 per window at 41 x 480 x 720, packed on the device) to rank 0, which merges them into the running result (WindowMerger: the
    blend is sequential in the window index, and round j holds exactly the next N windows) on a side stream WHILE every rank — rank 0 included —
    computes its window of round j + 1.  After the last round only that round's merge, the back-projection and the D2H copy remain (with one
    gather at the very end, as `run_windows` + `blend_and_merge_window_results` do, the whole merge is a serial tail on rank 0).
    Returns (rgb, disparity, poses, pointmaps) on rank 0 and None elsewhere; values are those of `blend_and_merge_window_results(run_windows(...),
    device=...)` bit for bit (same kernels, same order).  `timings` (a dict) receives 'windows_and_gather' and 'merge_tail' in seconds.
    Rank 0 never waits on the host for a merge: the per-pixel passes are enqueued on the side stream, each window's raymap comes to the host by a
    non-blocking copy and its camera algebra runs one round late (WindowMerger.add / _drain).
    `pinned=True`: the returned arrays are views of process-wide page-locked buffers — valid until the next merge of the same shape
    (`release_pinned_buffers()` frees them); copy them if they must live longer."""
    import time
    dist = _dist()
    rank, world = (dist.get_rank(), dist.get_world_size()) if dist is not None else (0, 1)
    collective = dist is not None and (world > 1 or force_collective)
    nccl = collective and dist.get_backend() == "nccl"
    if gather_device is not None:
        dev = torch.device(gather_device)
    else:
        dev = torch.device("cuda", torch.cuda.current_device()) if (nccl or (not collective and torch.cuda.is_available())) else torch.device("cpu")
    if collective and not nccl:
        dev = torch.device("cpu")
    on_gpu = dev.type == "cuda"
    side = torch.cuda.Stream(device=dev) if on_gpu else None
    merger, shapes, sizes = None, None, None
    n_rounds = (len(starts) + world - 1) // world
    t_begin = time.perf_counter()

    def merge(idx, rgb, disp, ray):
        nonlocal merger
        if merger is None:
            merger = WindowMerger(total_frames=starts[-1] + rgb.shape[0], window_frames=rgb.shape[0], frame_hw=tuple(disp.shape[1:]), height=height, width=width,
                                  device=dev, smooth_camera=smooth_camera, smooth_method=smooth_method, out_dtype=out_dtype, pinned=pinned)
        merger.add(WindowResult(starts[idx], rgb, disp, ray if ray.device.type == "cuda" else ray.cpu().numpy().copy()))

    for j in range(n_rounds):
        idx = j * world + rank
        out = call_window(starts[idx]) if idx < len(starts) else None
        parts = None if out is None else [_as_tensor(out.rgb), _as_tensor(out.disparity), _as_tensor(out.raymap)]
        if not collective:
            a, b, c = (t.to(dev) for t in parts)
            if side is not None:
                side.wait_stream(torch.cuda.current_stream(dev))
                with torch.cuda.stream(side):
                    merge(idx, a, b, c)
                for t in (a, b, c):
                    t.record_stream(side)
            else:
                merge(idx, a, b, c)
            continue
        if shapes is None:                                   # every rank has a window in round 0 unless there are fewer windows than ranks
            mine = tuple(tuple(t.shape) for t in parts) if parts is not None else None
            every = [None] * world
            dist.all_gather_object(every, mine)
            shapes = next(x for x in every if x is not None)
            sizes = [int(np.prod(x)) for x in shapes]
        payload = torch.empty(sum(sizes) + 1, dtype=torch.float32, device=dev)      # last element: 1 = slot holds a window
        if parts is not None:
            torch.cat([t.reshape(-1).to(dev, torch.float32) for t in parts] + [torch.ones(1, dtype=torch.float32, device=dev)], out=payload)
        else:
            payload.zero_()
        gathered = [torch.empty_like(payload) for _ in range(world)] if rank == 0 else None
        work = dist.gather(payload, gathered, dst=0, async_op=True)
        if rank != 0:
            work.wait()
            continue

        def merge_round(j=j, gathered=gathered, work=work):
            work.wait()                                      # stream-level under nccl: the merge kernels queue behind the gather
            for r, g in enumerate(gathered):
                i = j * world + r
                if i >= len(starts):
                    continue
                a, b, c, _valid = torch.split(g, sizes + [1])
                merge(i, a.view(shapes[0]), b.view(shapes[1]), c.view(shapes[2]))
        if side is not None:
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                merge_round()
            for g in gathered:
                g.record_stream(side)
        else:
            merge_round()
    if rank != 0:
        if timings is not None:
            timings["windows_and_gather"] = time.perf_counter() - t_begin
        return None
    if side is not None:
        torch.cuda.current_stream(dev).synchronize()         # this rank's last window is done (the merges of earlier rounds ran beside it)
    t_mid = time.perf_counter()
    if side is not None:
        torch.cuda.current_stream(dev).wait_stream(side)
    merged = merger.finish()
    if timings is not None:
        timings["windows_and_gather"] = t_mid - t_begin
        timings["merge_tail"] = time.perf_counter() - t_mid
    return merged


def blend_and_merge_window_results(results: Sequence[WindowResult], *, height: int, width: int, align_pointmaps: bool = False,
                                   smooth_camera: bool = True, smooth_method: str = "kalman", device: Optional[torch.device] = None,
                                   out_dtype=np.float64, pinned: bool = False):
    """The reference's sequential merge of overlapping windows (D:254-422), on the host in float64 like the reference:
    window k is brought into the frame of everything merged so far — disparity by a least-squares scale over the overlap
    (pixels with disparity > 0.1), camera poses by a similarity fitted on the overlapping cameras, focal lengths by their mean
    ratio — and cross-faded linearly over the overlap (poses by slerp); finally every frame is back-projected to a world-space
    point map.  Returns (rgb [N,H,W,3], disparity [N,H,W], poses [N,4,4], pointmaps [N,H,W,3]).
    Reference quirks kept on purpose: windows' raymaps are decoded in place (geometry.raymap_to_poses); the aligned poses of
    windows k >= 1 carry the similarity's scale in element [3,3] outside the overlap (apply_transformation on 4x4 inputs).
    `device`: run the per-pixel part (scale fit, cross-fades, back-projection: SURVEY.md §8f-2) as float64 torch operations on
    that device instead of numpy on the host (`WindowMerger`); the ≤ 41-pose camera algebra stays on the host either way.  `out_dtype` /
    `pinned` (device path only): see WindowMerger."""
    from . import geometry as G

    if device is not None and not align_pointmaps:
        return _merge_on_device(results, height=height, width=width, smooth_camera=smooth_camera, smooth_method=smooth_method,
                                device=torch.device(device), out_dtype=out_dtype, pinned=pinned)
    sm = smooth_method if smooth_camera else "none"
    first = results[0]
    n_win = first.rgb.shape[0]
    frame_shape = first.disparity.shape[1:]
    total = results[-1].start + results[-1].rgb.shape[0]
    # The reference re-allocates (np.ones) and re-copies every merged array for each window (O(windows²) traffic: ~8 GB of
    # float64 copies for a 192-frame clip); the same values are written here into arrays allocated once at the final length.
    rgb = np.empty((total, *frame_shape, 3))            # every element is written below
    disp = np.empty((total, *frame_shape))
    poses = np.empty((total, 4, 4))
    focals = np.empty((total,))
    pointmaps = np.empty((total, *frame_shape, 3)) if align_pointmaps else None
    rgb[:n_win], disp[:n_win] = first.rgb, first.disparity
    pm0 = G.postprocess_pointmap(first.disparity, first.raymap, vae_downsample_scale=8, ray_o_scale_inv=0.1,
                                 smooth_camera=smooth_camera, smooth_method=sm, with_pointmap=align_pointmaps)
    poses[:n_win] = pm0["camera_pose"]
    focals[:n_win] = (pm0["intrinsics"][:, 0, 0] + pm0["intrinsics"][:, 1, 1]) / 2
    if align_pointmaps:
        pointmaps[:n_win] = pm0["pointmap"]
    end = n_win                                                             # frames merged so far

    for k in range(1, len(results)):
        r, t0 = results[k], results[k].start
        t1 = t0 + r.rgb.shape[0]
        ov = results[k - 1].start + n_win - t0                              # frames shared with what is merged so far
        assert end == t0 + ov
        fade = np.linspace(1, 0, ov)                                        # weight of the already merged frames

        # disparity: scale onto the merged overlap, cross-fade
        cols = frame_shape[-1]
        scale = G.compute_scale(r.disparity[:ov].reshape(1, -1, cols), disp[t0:end].reshape(1, -1, cols),
                                (r.disparity[:ov].reshape(1, -1, cols) > 0.1))
        w_disp = scale * r.disparity
        disp[t0:end] = disp[t0:end] * fade[:, None, None] + w_disp[:ov] * (1 - fade[:, None, None])
        disp[end:t1] = w_disp[ov:]

        # colour: cross-fade
        rgb[t0:end] = rgb[t0:end] * fade[:, None, None, None] + r.rgb[:ov] * (1 - fade[:, None, None, None])
        rgb[end:t1] = r.rgb[ov:]

        # cameras: similarity from the overlapping cameras, slerp / lerp inside the overlap
        w_poses, fov_x, fov_y = G.raymap_to_poses(r.raymap, ray_o_scale_inv=0.1)
        aR, aT, aS = G.align_camera_extrinsics(w_poses[:ov], poses[t0:end])
        w_aligned = G.apply_transformation(w_poses, aR, aT, aS)
        for t in range(ov):
            poses[t0 + t] = G.interpolate_poses(poses[t0 + t], w_aligned[t], fade[t])
        poses[end:t1] = w_aligned[ov:]

        # focal lengths: mean ratio over the overlap, cross-fade
        w_focals = G.focals_from_fov(w_poses.shape[0], r.disparity.shape[1], r.disparity.shape[2], fov_x, fov_y)
        w_focals = (focals[t0:end] / w_focals[:ov]).mean() * w_focals
        focals[t0:end] = focals[t0:end] * fade + w_focals[:ov] * (1 - fade)
        focals[end:t1] = w_focals[ov:]

        if align_pointmaps:
            w_pm = G.postprocess_pointmap(disp[t0:t1], r.raymap, vae_downsample_scale=8, camera_pose=w_aligned, focal=w_focals,
                                          ray_o_scale_inv=0.1, smooth_camera=smooth_camera, smooth_method=sm)["pointmap"]
            pointmaps[t0:end] = pointmaps[t0:end] * fade[:, None, None, None] + w_pm[:ov] * (1 - fade[:, None, None, None])
            pointmaps[end:t1] = w_pm[ov:]
        end = t1

    if not align_pointmaps:
        pointmaps = np.empty((total, *frame_shape, 3))
        for i, f in enumerate(focals):
            pointmaps[i] = G.project(1 / np.clip(disp[i], 1e-8, 1e8), np.array([[f, 0, 0.5 * width], [0, f, 0.5 * height], [0, 0, 1]]),
                                     poses[i])
    return rgb, disp, poses, pointmaps


_PINNED: dict = {}


def release_pinned_buffers() -> None:
    """Free the page-locked result buffers of `pinned=True` merges (1.9 GB for a 192-frame float32 clip, 3.7 GB in float64).  Arrays returned by
    earlier `pinned=True` merges are VIEWS of these buffers: they are overwritten by the next merge of the same shape and must not be used after
    this call — copy what has to outlive either."""
    _PINNED.clear()


def _pinned(tag: str, shape, dtype) -> torch.Tensor:
    """Page-locked host buffers for the merged arrays, allocated once per (shape, dtype) and reused by later merges: a D2H copy into pageable
    memory runs at ~7 GB/s on this box (0.5 s for the 3.7 GB a 192-frame clip's float64 arrays), into pinned memory at PCIe speed."""
    key = (tag, tuple(shape), dtype)         # rgb and the point maps have the same shape: one buffer per ROLE
    if key not in _PINNED:
        _PINNED[key] = torch.empty(shape, dtype=dtype, pin_memory=True)
    return _PINNED[key]


class WindowMerger:
    """The per-pixel work of blend_and_merge_window_results (align_pointmaps=False) on `device`, INCREMENTALLY: `add(result)` merges the next
    window (windows must arrive in start order — the blend is sequential in the window index: D:269-401), `finish()` back-projects and copies
    the merged arrays to the host.  Same operations, dtypes and order as the host path (float32 operands of the scale fit, float64 everything
    else).  On an MI355X the three passes are HIP kernels of libaether_hip.so (csrc/merge_kernels.hip: masked scale-fit reduction, fused
    scale + cross-fade of disparity and colour, back-projection) reading the gathered fp32 window outputs where `run_windows` left them; on any
    other device the same arithmetic runs as torch operations (CPU tests).  One deliberate difference: the HIP scale fit accumulates Σ m·p·t
    and Σ m·p² in float64 and divides in float64 before rounding the scale to float32, where the reference (compute_scale, U:847-864) sums and
    divides in float32 — the kernel's value is the more accurate one and differs from the float32 result by ~1e-7 relative (the tests'
    tolerance against the reference's own outputs, tests/golden/blend.npz, is 1e-5).
    `out_dtype`: dtype of the returned rgb / disparity / pointmap arrays — float64 like the reference's (np.ones defaults, D:262-268) unless
    the caller asks for float32 (half the D2H bytes; scripts/demo.py does unless --float64_outputs: everything it writes is uint8 / float32).
    `pinned`: copy into reusable page-locked buffers (the returned arrays are views of them: valid until the next merge of the same shape)."""

    def __init__(self, *, total_frames: int, window_frames: int, frame_hw, height: int, width: int, device, smooth_camera: bool = True,
                 smooth_method: str = "kalman", out_dtype=np.float64, pinned: bool = False):
        self.device = torch.device(device)
        self.total, self.n_win, (self.H, self.W) = int(total_frames), int(window_frames), frame_hw
        self.height, self.width = height, width
        self.smooth_camera, self.sm = smooth_camera, (smooth_method if smooth_camera else "none")
        self.out_dtype, self.pinned = np.dtype(out_dtype), pinned
        self.native = self.device.type == "cuda"
        f64 = dict(dtype=torch.float64, device=self.device)
        if self.native:
            from . import _lib
            self._lib, self.lib = _lib, _lib.load()
            self.scratch = torch.empty(4096 + 3, **f64)
        self.rgb = torch.empty((self.total, self.H, self.W, 3), **f64)
        self.disp = torch.empty((self.total, self.H, self.W), **f64)
        self.poses = np.empty((self.total, 4, 4))
        self.focals = np.empty((self.total,))
        self.end, self.prev_start, self.count = 0, None, 0
        # camera algebra of the windows whose raymap is still on its way to the host: (raymap pinned buffer | array, event | None, t0, ov, first)
        self._pending: list = []
        self._free_ray: list = []

    def _up(self, a):
        return (a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))).to(self.device)

    def _up32(self, a):
        return self._up(a).to(torch.float32).contiguous()

    def _write_window(self, r_rgb, r_disp, t0, ov, fade_h, scale_dev):
        """frames [t0, t0 + n) of the merged arrays from one window (cross-fade over the first `ov`)."""
        n = r_rgb.shape[0]
        fade_c = (C.c_double * max(ov, 1))(*fade_h) if ov else None
        stream = torch.cuda.current_stream(self.device).cuda_stream
        self._lib.check(self.lib.aether_merge_window(r_rgb.data_ptr(), r_disp.data_ptr(), self.rgb[t0].data_ptr(), self.disp[t0].data_ptr(), n, ov,
                                                     self.H * self.W, fade_c, None if scale_dev is None else scale_dev.data_ptr(), stream),
                        "aether_merge_window")

    def add(self, r: "WindowResult") -> None:
        """Merge the next window.  The per-pixel passes are enqueued on the current stream without any host synchronisation; the camera algebra
        (a few dozen 4x4 matrices, host numpy like the reference's) needs the window's raymap on the host: a device raymap is fetched by a
        non-blocking copy into page-locked memory + an event, and the algebra of a window runs when that event has fired — at a later `add` or in
        `finish` — so the caller can enqueue its next window before this one's merge has executed (`run_windows_merged`).  Order of the host
        arithmetic, hence every value, is that of the immediate form."""
        H, W, n_win, device = self.H, self.W, self.n_win, self.device
        rgb, disp = self.rgb, self.disp
        if self.count == 0:
            assert r.start == 0
            if self.native:
                self._write_window(self._up32(r.rgb), self._up32(r.disparity), 0, 0, [], None)
            else:
                rgb[:n_win], disp[:n_win] = self._up(r.rgb), self._up(r.disparity)
            self._queue_cameras(r.raymap, 0, 0, True)
            self.end, self.prev_start, self.count = n_win, 0, 1
            self._drain(block=False)
            return
        t0, end = r.start, self.end
        t1 = t0 + r.rgb.shape[0]
        ov = self.prev_start + n_win - t0
        assert end == t0 + ov
        fade_h = np.linspace(1, 0, ov)
        # the HIP merge kernels take 1..64 overlapping frames; windows that do not overlap (the reference's compute_scale then returns
        # 0, U:847-864) or overlap by more run the same arithmetic as torch operations on the device
        if self.native and 0 < ov <= 64:
            r_disp, r_rgb = self._up32(r.disparity), self._up32(r.rgb)
            stream = torch.cuda.current_stream(device).cuda_stream
            # scale fit (U:847-864) -> device scalar, then ONE pass: scale, cross-fade of disparity and colour, tail frames
            self._lib.check(self.lib.aether_merge_scale_fit(r_disp.data_ptr(), disp[t0].data_ptr(), ov * H * W, self.scratch.data_ptr(), 4096,
                                                            self.scratch[4096:].data_ptr(), stream), "aether_merge_scale_fit")
            self._write_window(r_rgb, r_disp, t0, ov, fade_h.tolist(), self.scratch[4098:])
        else:
            fade = torch.from_numpy(fade_h).to(device)
            r_disp, r_rgb = self._up(r.disparity), self._up(r.rgb)
            # scale fit: float32 operands and float32 sums like the reference's torch code (U:847-864)
            p, t = r_disp[:ov].float(), disp[t0:end].float()
            m = (p > 0.1).float()
            den = float((m * p * p).sum())
            scale = float((m * p * t).sum()) / den if den != 0 else 0.0
            w_disp = scale * r_disp
            disp[t0:end] = disp[t0:end] * fade[:, None, None] + w_disp[:ov] * (1 - fade[:, None, None])
            disp[end:t1] = w_disp[ov:]
            rgb[t0:end] = rgb[t0:end] * fade[:, None, None, None] + r_rgb[:ov] * (1 - fade[:, None, None, None])
            rgb[end:t1] = r_rgb[ov:]
        self._queue_cameras(r.raymap, t0, ov, False)
        self.end, self.prev_start, self.count = t1, t0, self.count + 1
        self._drain(block=False)

    def _queue_cameras(self, raymap, t0: int, ov: int, first: bool) -> None:
        if isinstance(raymap, torch.Tensor) and raymap.device.type == "cuda":
            buf = self._free_ray.pop() if (self._free_ray and self._free_ray[-1].shape == raymap.shape) else torch.empty(raymap.shape, dtype=torch.float32, pin_memory=True)
            buf.copy_(raymap, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(raymap.device))
            self._pending.append((buf, ev, t0, ov, first))
        else:
            self._pending.append((_host(raymap), None, t0, ov, first))

    def _drain(self, block: bool) -> None:
        """Camera algebra of the queued windows, in window order, for every window whose raymap has arrived (all of them when `block`)."""
        while self._pending:
            buf, ev, t0, ov, first = self._pending[0]
            if ev is not None:
                if block:
                    ev.synchronize()
                elif not ev.query():
                    return
            self._pending.pop(0)
            if ev is not None:
                ray = buf.numpy().copy()                     # raymap_to_poses decodes in place: work on a copy, recycle the pinned buffer
                self._free_ray.append(buf)
            else:
                ray = buf
            self._cameras(ray, t0, ov, first)

    def _cameras(self, raymap: np.ndarray, t0: int, ov: int, first: bool) -> None:
        """Cameras and focal lengths of one window (D:292-401 without the per-pixel part): host numpy, float64 like the reference."""
        from . import geometry as G
        H, W, n_win = self.H, self.W, self.n_win
        poses, focals = self.poses, self.focals
        if first:
            # with_pointmap=False reads only the raymap (U:283-351): no disparity crosses to the host
            pm0 = G.postprocess_pointmap(None, raymap, vae_downsample_scale=8, ray_o_scale_inv=0.1,
                                         smooth_camera=self.smooth_camera, smooth_method=self.sm, with_pointmap=False)
            poses[:n_win] = pm0["camera_pose"]
            focals[:n_win] = (pm0["intrinsics"][:, 0, 0] + pm0["intrinsics"][:, 1, 1]) / 2
            return
        end = t0 + ov
        fade_h = np.linspace(1, 0, ov)
        w_poses, fov_x, fov_y = G.raymap_to_poses(raymap, ray_o_scale_inv=0.1)
        t1 = t0 + w_poses.shape[0]
        aR, aT, aS = G.align_camera_extrinsics(w_poses[:ov], poses[t0:end])
        w_aligned = G.apply_transformation(w_poses, aR, aT, aS)
        for i in range(ov):
            poses[t0 + i] = G.interpolate_poses(poses[t0 + i], w_aligned[i], fade_h[i])
        poses[end:t1] = w_aligned[ov:]
        w_focals = G.focals_from_fov(w_poses.shape[0], H, W, fov_x, fov_y)
        w_focals = (focals[t0:end] / w_focals[:ov]).mean() * w_focals
        focals[t0:end] = focals[t0:end] * fade_h + w_focals[:ov] * (1 - fade_h)
        focals[end:t1] = w_focals[ov:]

    def finish(self):
        """Back-projection (U:393-403): world = pose[:3,:4] · [K⁻¹ · (u+.5, v+.5, 1) · depth ; 1], pixel grid in float32 like the reference; then
        the D2H copies (rgb and disparity leave on a copy stream while the back-projection kernel runs)."""
        assert self.end == self.total, "windows missing"
        self._drain(block=True)
        device, total, H, W = self.device, self.total, self.H, self.W
        K = np.zeros((total, 3, 3))
        K[:, 0, 0] = K[:, 1, 1] = self.focals
        K[:, 0, 2], K[:, 1, 2], K[:, 2, 2] = 0.5 * self.width, 0.5 * self.height, 1.0
        K_inv = torch.from_numpy(np.linalg.inv(K)).to(device)
        P = torch.from_numpy(self.poses[:, :3, :4].copy()).to(device)
        tdt = torch.float64 if self.out_dtype == np.float64 else torch.float32

        def to_host(tag, t, copy_stream=None):
            t = t if t.dtype == tdt else t.to(tdt)
            if not (self.native and self.pinned):
                return t.cpu()
            dst = _pinned(tag, t.shape, tdt)
            if copy_stream is None:
                dst.copy_(t, non_blocking=True)
            else:
                copy_stream.wait_stream(torch.cuda.current_stream(device))
                with torch.cuda.stream(copy_stream):
                    dst.copy_(t, non_blocking=True)
                t.record_stream(copy_stream)
            return dst

        side = torch.cuda.Stream(device=device) if (self.native and self.pinned) else None
        rgb_h, disp_h = to_host("rgb", self.rgb, side), to_host("disparity", self.disp, side)
        pointmaps = torch.empty((total, H, W, 3), dtype=torch.float64, device=device)
        if self.native:
            self._lib.check(self.lib.aether_backproject(self.disp.data_ptr(), K_inv.contiguous().data_ptr(), P.contiguous().data_ptr(), pointmaps.data_ptr(),
                                                        total, H, W, torch.cuda.current_stream(device).cuda_stream), "aether_backproject")
        else:
            v, u = torch.meshgrid(torch.arange(H, device=device), torch.arange(W, device=device), indexing="ij")
            pix = torch.stack([u.reshape(-1) + 0.5, v.reshape(-1) + 0.5, torch.ones(H * W, device=device)], 0).float().double()   # [3, HW]
            step = 16
            for i in range(0, total, step):
                j = min(i + step, total)
                depth = (1 / self.disp[i:j].clamp(1e-8, 1e8)).reshape(j - i, 1, H * W)
                cam = (K_inv[i:j] @ pix) * depth                                                         # [n, 3, HW]
                world = P[i:j, :, :3] @ cam + P[i:j, :, 3:]                                               # [n, 3, HW]
                pointmaps[i:j] = world.transpose(1, 2).reshape(j - i, H, W, 3)
        pm_h = to_host("pointmaps", pointmaps)
        if self.native:
            if side is not None:
                torch.cuda.current_stream(device).wait_stream(side)
            torch.cuda.current_stream(device).synchronize()
        return rgb_h.numpy(), disp_h.numpy(), self.poses, pm_h.numpy()


def _merge_on_device(results: Sequence[WindowResult], *, height: int, width: int, smooth_camera: bool, smooth_method: str,
                     device: torch.device, out_dtype=np.float64, pinned: bool = False):
    """blend_and_merge_window_results(device=...): every window through a WindowMerger, in order."""
    first = results[0]
    m = WindowMerger(total_frames=results[-1].start + results[-1].rgb.shape[0], window_frames=first.rgb.shape[0], frame_hw=tuple(first.disparity.shape[1:]),
                     height=height, width=width, device=device, smooth_camera=smooth_camera, smooth_method=smooth_method, out_dtype=out_dtype, pinned=pinned)
    for r in results:
        m.add(r)
    return m.finish()


def _host(a):
    return a.cpu().numpy() if isinstance(a, torch.Tensor) else a


def blend_rgb(results: Sequence[WindowResult], total_frames: int) -> np.ndarray:
    """Linear cross-fade of the RGB frames of overlapping windows (the colour part of D:254-422; the geometric part —
    disparity scale fitting, pose alignment — stays with the reference's numpy post-processing)."""
    f, h, w, c = results[0].rgb.shape
    out = np.zeros((total_frames, h, w, c), np.float32)
    filled = 0
    for r in results:
        s = r.start
        overlap = max(filled - s, 0)
        if overlap > 0:
            wgt = np.linspace(0.0, 1.0, overlap, dtype=np.float32)[:, None, None, None]
            out[s:s + overlap] = out[s:s + overlap] * (1 - wgt) + r.rgb[:overlap] * wgt
        out[s + overlap:s + f] = r.rgb[overlap:]
        filled = max(filled, s + f)
    return out
