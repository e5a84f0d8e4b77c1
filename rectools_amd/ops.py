"""torch.autograd bindings of the HIP kernels (C ABI: include/rectools_hip.h).

PyTorch is plumbing here: it owns the buffers (caching allocator), the stream and the autograd tape; every
forward/backward below is one or a few `rt_*` launches on torch's current HIP stream.  There is no eager /
CPU fallback: tensors must be fp32, contiguous where stated, and on a HIP device.

All activations are 2-D `[M, d]` (M = batch * session_max_len).
"""
from __future__ import annotations

import math
import os
import typing as tp

import torch

from . import _lib

ACT_NONE, ACT_RELU, ACT_GELU, ACT_SILU, ACT_SIGMOID = 0, 1, 2, 3, 4
LOSS_BCE, LOSS_GBCE, LOSS_SAMPLED_SOFTMAX = 0, 1, 2


# Optional per-call HIP-event instrumentation (bench.py's roofline pass): name -> [(start, stop, tag), ...]
_TIMING: tp.Optional[tp.Dict[str, tp.List[tp.Tuple[torch.cuda.Event, torch.cuda.Event, tp.Any]]]] = None


_TIMING_SINGLE_STREAM = True


def start_timing(single_stream: bool = True) -> None:
    """Record a HIP event pair around every `rt_*` launch, on the stream it is launched on.  single_stream=True also
    routes the weight-gradient products to the main stream (undisturbed kernel durations); False keeps the side stream
    of the product configuration (durations then include the overlap with the main stream's kernels)."""
    global _TIMING, _TIMING_SINGLE_STREAM
    _TIMING = {}
    _TIMING_SINGLE_STREAM = single_stream
    _lib.load().rt_timing_enable(2 if single_stream else 1)     # the native block executor brackets its internal launches itself


def stop_timing() -> tp.Dict[str, tp.List[tp.Tuple[float, tp.Any]]]:
    """-> {kernel entry point: [(milliseconds, tag), ...]} for every call since start_timing()."""
    global _TIMING
    rec, _TIMING = _TIMING or {}, None
    torch.cuda.synchronize()
    out = {k: [(a.elapsed_time(b), tag) for a, b, tag in v] for k, v in rec.items()}
    out.pop("rt_sasrec_block_packed_fwd", None)      # the executor's own records (below) itemise these calls
    out.pop("rt_sasrec_block_packed_bwd", None)
    import ctypes

    lib = _lib.load()
    cap = 1 << 16
    ids, ms, tags, n = (ctypes.c_int32 * cap)(), (ctypes.c_float * cap)(), (ctypes.c_int64 * (3 * cap))(), ctypes.c_int32(0)
    lib.rt_timing_collect(ids, ms, tags, cap, ctypes.byref(n))
    lib.rt_timing_enable(0)
    for i in range(n.value):
        name = _NATIVE_TIMING_NAMES.get(ids[i], "rt_misc")
        out.setdefault(name, []).append((float(ms[i]), (int(tags[3 * i]), int(tags[3 * i + 1]), int(tags[3 * i + 2]))))
    return out


_NATIVE_TIMING_NAMES = {0: "rt_gemm", 1: "rt_gemm_grouped", 2: "rt_layernorm_fwd", 3: "rt_layernorm_bwd_fused", 4: "rt_act_dropout_fwd",
                        5: "rt_act_dropout_bwd", 6: "rt_mha_varlen_train_fwd", 7: "rt_mha_varlen_bwd", 8: "rt_mha_varlen_last_fwd",
                        9: "rt_misc", 10: "rt_mha_varlen_bidir_fwd", 11: "rt_mha_varlen_bidir_bwd", 12: "rt_ffn_fused_fwd", 13: "rt_ffn_fused_bwd"}


_FN: tp.Dict[str, tp.Any] = {}   # bound C entry points (one getattr per name instead of one per launch)


def _c(name: str, *args: tp.Any, tag: tp.Any = None, stream: tp.Optional[int] = None, timed_as: tp.Optional[str] = None) -> None:
    """Call `rt_<name>(*args, stream)`; tensors are passed as raw device pointers.  stream: a raw HIP stream handle (the library's
    side stream, `_native_side_fork`); default torch's current stream."""
    fn = _FN.get(name)
    if fn is None:
        fn = _FN[name] = getattr(_lib.load(), name)
    conv = [a.data_ptr() if isinstance(a, torch.Tensor) else a for a in args]
    if stream is not None:
        status = fn(*conv, stream)
    elif _TIMING is None:
        status = fn(*conv, _lib.current_stream())
    else:  # events on the stream the kernel is launched on (torch's current stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        status = fn(*conv, _lib.current_stream())
        e1.record()
        _TIMING.setdefault(timed_as or name, []).append((e0, e1, tag))
    _lib.check(status, name)


def _native_side_fork() -> tp.Optional[int]:
    """Fork the library's side stream off torch's current stream (`rt_side_fork`): -> its raw handle, or None when the side stream
    is off (RT_SIDE_STREAM=0, single-stream instrumentation).  Launch optimiser-only work on it with `_c(..., stream=handle)`, keep
    every tensor it touches in `_NATIVE_KEEPALIVE`; `join_side_streams()` (end of backward / FlatAdam.step) joins it."""
    import ctypes

    if not _side_enabled():
        return None
    h = ctypes.c_void_p()
    _lib.check(_lib.load().rt_side_fork(_lib.current_stream(), ctypes.byref(h)), "rt_side_fork")
    if not h.value:
        return None
    if not _NATIVE_KEEPALIVE:
        torch.autograd.Variable._execution_engine.queue_callback(join_side_streams)
    return h.value


_PREP_KEEPALIVE: tp.List[tp.Any] = []     # buffers of work issued on the side stream during the FORWARD pass (see `_side_fork_forward`)
_PREP_STEP = -1                            # RNG.step of the forward pass that filled it


def _side_fork_forward() -> tp.Optional[int]:
    """`_native_side_fork` for work issued during the forward pass — the counting sorts that depend on the batch's ids alone
    (`prepare_sampled_pairs`, `_EmbedPacked.forward`).  No autograd callback can be queued there: the consumer joins
    (`join_side_streams()` in front of the kernel that reads the result, unless it runs on the side stream itself)."""
    import ctypes

    if not _side_enabled():
        return None
    h = ctypes.c_void_p()
    _lib.check(_lib.load().rt_side_fork(_lib.current_stream(), ctypes.byref(h)), "rt_side_fork")
    return h.value or None


def _chk(t: torch.Tensor, what: str) -> torch.Tensor:
    if not t.is_cuda or t.dtype != torch.float32:
        raise _lib.HipLibraryError(f"{what}: expected a float32 HIP tensor (no CPU fallback), got {t.dtype} on {t.device}")
    return t


class DropoutRng:
    """Counter-based dropout streams: (seed, stream_id) pairs; the backward kernels regenerate the masks."""

    def __init__(self, seed: int = 0) -> None:
        self.seed = seed & 0xFFFFFFFFFFFFFFFF
        self.step = 0
        self._stream = 0

    def next_step(self) -> None:
        self.step += 1
        self._stream = 0

    def next(self) -> tp.Tuple[int, int]:
        self._stream += 1
        return (self.seed + 0x9E3779B97F4A7C15 * self.step) & 0xFFFFFFFFFFFFFFFF, self._stream


RNG = DropoutRng(0)


# --------------------------------------------------------------------------------------------------
# side stream for weight gradients
# --------------------------------------------------------------------------------------------------
# dW / db are needed only by the optimiser, while the data-gradient chain is the critical path of the backward pass.
# Each exact-tile GEMM of the step fills 1.56 "rounds" of the 256 CUs (78% of the machine); issuing the wgrad products
# on a second HIP stream lets their workgroups occupy the CUs that the dgrad products, the attention kernels and the
# memory-bound row kernels leave idle.  The main stream re-joins (`join_side_streams`) in a callback the autograd
# engine runs at the end of the backward pass.  RT_SIDE_STREAM=0 disables the second stream.
_SIDE: tp.Dict[torch.device, "torch.cuda.Stream"] = {}
_SIDE_DIRTY: tp.Set[torch.device] = set()


def _side_enabled() -> bool:
    import os

    return os.environ.get("RT_SIDE_STREAM", "1") != "0" and (_TIMING is None or not _TIMING_SINGLE_STREAM)


def _steals_grad(*params: tp.Optional[torch.Tensor]) -> bool:
    """True when autograd will simply ADOPT the gradient tensors returned for these parameters (leaf tensors without an
    existing `.grad`): no kernel touches them before the end-of-backward join, so they may still be in flight on the
    side stream.  Slices / views of parameters, or gradient accumulation into an existing `.grad`, run autograd kernels
    on the main stream right after the node returns — then the node has to join before returning."""
    return all(p is None or (p.is_leaf and p.grad is None) for p in params)


_DIRTY_STREAM: tp.Dict[torch.device, "torch.cuda.Stream"] = {}     # the stream `_OnSide` used on a device of `_SIDE_DIRTY`


def shared_side_stream(dev: torch.device) -> "torch.cuda.Stream":
    """THE weight-gradient queue of a device in this process, as a torch stream: the library's own side stream (`rt_side_stream`) wrapped as
    an ExternalStream — the Python-issued weight gradients (`_OnSide`) go where the native executors' go, instead of owning a second
    side stream (measured equal on the BERT4Rec and HSTU steps, whose nodes use both; one stream less in a process whose loops also own
    a prefetch stream: with more streams than hardware queues two of them share one, decided at run time — the HSTU loop measured
    20.3 k seqs/s in one default bench line and 24.1 k in the next).  The loop's next-batch collate keeps its own stream: behind the weight
    gradients on this one it stands between them and the join in front of Adam (BERT4Rec 53.4 -> 50.7 k seqs/s, C2 - 0.4 %).
    A plain torch stream when the library's is switched off (RT_SIDE_STREAM=0)."""
    import ctypes

    dev = torch.device(dev)
    side = _SIDE.get(dev)
    if side is None:
        h = ctypes.c_void_p()
        with torch.cuda.device(dev):
            _lib.check(_lib.load().rt_side_stream(ctypes.byref(h)), "rt_side_stream")
        side = _SIDE[dev] = torch.cuda.ExternalStream(h.value, device=dev) if h.value else torch.cuda.Stream(device=dev)
    return side


class _OnSide:
    """Context: run the enclosed launches on the side stream, ordered after everything issued so far on the current one.
    Tensors touched inside must be passed to `uses()` so that the caching allocator does not recycle them early.
    defer=True leaves the results in flight until the end of the backward pass; defer=False requires `join_now()`
    before the autograd node returns."""

    def __init__(self, dev: torch.device, defer: bool = True) -> None:
        self.dev = dev
        self.defer = defer
        self.enabled = _side_enabled()

    def __enter__(self) -> "_OnSide":
        if self.enabled:
            side = shared_side_stream(self.dev)
            side.wait_event(torch.cuda.current_stream(self.dev).record_event())
            self.side = side
            self.ctx = torch.cuda.stream(side)
            self.ctx.__enter__()
            if self.defer:
                if not _SIDE_DIRTY:  # join when the autograd engine has issued the whole backward pass
                    torch.autograd.Variable._execution_engine.queue_callback(join_side_streams)
                _SIDE_DIRTY.add(self.dev)
                _DIRTY_STREAM[self.dev] = side
        return self

    def uses(self, *tensors: torch.Tensor) -> None:
        if self.enabled:
            for t in tensors:
                t.record_stream(self.side)

    def __exit__(self, *exc: tp.Any) -> None:
        if self.enabled:
            self.ctx.__exit__(*exc)

    def join_now(self) -> None:
        if self.enabled and not self.defer:
            torch.cuda.current_stream(self.dev).wait_event(self.side.record_event())


_NATIVE_KEEPALIVE: tp.List[tp.Any] = []   # buffers the native executor's side stream may still be reading (until the next join)


def join_side_streams() -> None:
    """Main stream waits for every weight-gradient product issued on a side stream — torch's (the Python autograd nodes) and the
    library's own (the native block executor, csrc/rt_block.hip)."""
    for dev in list(_SIDE_DIRTY):
        torch.cuda.current_stream(dev).wait_event(_DIRTY_STREAM[dev].record_event())
    _SIDE_DIRTY.clear()
    if _NATIVE_KEEPALIVE or _PREP_KEEPALIVE:
        _c("rt_side_join")
        _NATIVE_KEEPALIVE.clear()
        _PREP_KEEPALIVE.clear()
    _TABLE_GRAD_ON_SIDE.clear()


def side_streams_reach(stream: "torch.cuda.Stream") -> None:
    """`stream` waits for everything issued so far on the weight-gradient side streams; nothing is joined or released (the step's
    `join_side_streams` still does that).  For a reader of the block weights' gradients that is not the main stream: the early
    gradient exchange of a data-parallel step (`lightning.FlatAdam.begin_early_exchange`)."""
    for dev in list(_SIDE_DIRTY):
        stream.wait_event(_DIRTY_STREAM[dev].record_event())
    if _NATIVE_KEEPALIVE or _PREP_KEEPALIVE:
        _lib.check(_lib.load().rt_side_reach(stream.cuda_stream), "rt_side_reach")


# --------------------------------------------------------------------------------------------------
# dense
# --------------------------------------------------------------------------------------------------
_GEMM_WS: tp.Dict[tp.Tuple[torch.device, int], torch.Tensor] = {}


def _gemm(A, lda, a_kc, B, ldb, b_kc, C, ldc, bias, R, ldr, M, N, K, relu=0, split_k=1, a_rowsum=None) -> None:
    ws, ws_bytes = None, 0
    if split_k > 1:
        ws_bytes = _lib.load().rt_gemm_workspace_bytes(M, N, K, split_k)
        key = (C.device, _lib.current_stream())
        ws = _GEMM_WS.get(key)
        if ws is None or ws.numel() < ws_bytes:  # one growing scratch per (device, stream): stream-ordered reuse
            ws = _GEMM_WS[key] = torch.empty((max(ws_bytes, 1 << 24),), dtype=torch.uint8, device=C.device)
    _c("rt_gemm", A, lda, a_kc, B, ldb, b_kc, C, ldc, bias, R, ldr, a_rowsum, M, N, K, relu, split_k, ws, ws_bytes,
       tag=(M, N, K))


def _gemm_group(problems: tp.Sequence[tp.Tuple], a_kc: int, b_kc: int) -> None:
    """ONE launch for up to 4 independent products (A, lda, B, ldb, C, ldc, bias, R, ldr, M, N, K, relu) of the same operand
    layouts (`rt_gemm_grouped`): the tail tiles of one product run next to the head tiles of the next."""
    import ctypes

    arr = (_lib.GemmProblem * len(problems))()
    ptr = lambda t: None if t is None else t.data_ptr()   # noqa: E731
    for q, (A, lda, B, ldb, C, ldc, bias, R, ldr, M, N, K, relu) in zip(arr, problems):
        q.A, q.lda, q.B, q.ldb, q.C, q.ldc = ptr(A), lda, ptr(B), ldb, ptr(C), ldc
        q.bias, q.R, q.ldr, q.M, q.N, q.K, q.relu = ptr(bias), ptr(R), ldr, M, N, K, relu
    _c("rt_gemm_grouped", ctypes.cast(arr, ctypes.c_void_p), len(problems), a_kc, b_kc,
       tag=(sum(p[9] * p[10] * p[11] for p in problems), 1, 1))


_WGRAD_SPLIT_CAP = 64   # upper bound of the wgrad split-K factor


def _wgrad_splits(k_rows: int, out_rows: int = 0, out_cols: int = 0) -> int:
    """Split-K factor of a weight-gradient product dW [out_rows, out_cols] = sum over k_rows.  Small outputs (256 x 256 = 4
    tiles at C2) need every slice they can get to fill 256 CUs; large ones (eSASRec's 2048 x 512 = 64 tiles) were cut into 64
    slices of 12 k-steps each, 4096 workgroups whose slab writes and 64-way combine cost more than the parallelism bought:
    aim at ~6 workgroups per CU."""
    sp = max(1, min(_WGRAD_SPLIT_CAP, k_rows // 384))
    if out_rows > 0 and out_cols > 0:
        tiles = ((out_rows + 127) // 128) * ((out_cols + 127) // 128)
        sp = max(1, min(sp, -(-1536 // tiles)))
    return sp


def _deep_k_splits(M: int, N: int, K: int) -> int:
    """Split-K factor of a product with a small output and a deep reduction (the dS = G E product of the full-softmax loss:
    3,840 x 256 outputs = 60 tiles, K = 26,752 — unsplit it ran on 60 of 256 CUs for 1.7 ms).  Aim at ~3 workgroups per CU
    with slices of at least 512."""
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    if tiles >= 256 or K < 2048:
        return 1
    return max(1, min(K // 512, -(-768 // tiles), 64))


# ---- products against a weight whose bf16 planes are at hand (K7w outside the native executors) ---------------------------------------
# A layer stack arms its planes around its forward pass (`active_planes`); the autograd nodes below look a weight up in them
# (`WeightPlanes.of`) and keep what they found for their backward pass — the planes stay what they are until the stack's next forward
# pass re-splits them, and the weights do not change between a forward pass and its backward pass.
_ACTIVE_PLANES: tp.Optional["WeightPlanes"] = None


class active_planes:      # pylint: disable=invalid-name
    """`with ops.active_planes(planes):` — products of `ops.linear / matmul_nn / stu_layer*` inside read weights of `planes`' range from
    the pre-split planes (`rt_gemm_wp`).  None = plain `rt_gemm` (also what every product outside such a block is)."""

    def __init__(self, planes: tp.Optional["WeightPlanes"]) -> None:
        self.planes = planes if planes is not None and planes.ok and weight_planes_enabled() else None

    def __enter__(self) -> "active_planes":
        global _ACTIVE_PLANES
        self.prev, _ACTIVE_PLANES = _ACTIVE_PLANES, self.planes
        return self

    def __exit__(self, *exc: tp.Any) -> None:
        global _ACTIVE_PLANES
        _ACTIVE_PLANES = self.prev


def _planes_of(w: torch.Tensor) -> tp.Optional[tp.Tuple[int, int, tp.Any]]:
    """(plane-0 pointer, plane stride, keep-alive) of a weight inside the armed planes, else None."""
    pl = _ACTIVE_PLANES
    if pl is None or not w.is_contiguous():
        return None
    ptr = pl.of(w)
    return None if ptr is None else (ptr, pl.stride, pl.planes)


def _gemm_w(A, lda, W, ldw, w_kc, wp, C, ldc, bias, R, ldr, M, N, K, relu=0) -> None:
    """C[M, N] = A[M, K] . W' (+ bias) (+ R) (relu), A rows K-contiguous; W' = W[N, K] (w_kc = 1) or W[K, N] (w_kc = 0).  With `wp`
    (`_planes_of(W)`) and an exact tile grid in N and K the rows of the full 128-row tiles run on `rt_gemm_wp` (the weight's split read
    from the planes: half the split arithmetic of `rt_gemm`'s loop, the same six bf16 products per fp32 product), the rows behind them —
    and every product without planes — on `rt_gemm`."""
    m0 = 0
    if wp is not None and wp[0] % 16 == 0 and M >= 128 and N % 128 == 0 and K % 32 == 0 and lda % 4 == 0 and ldw % 8 == 0 and A.data_ptr() % 16 == 0 \
            and C.data_ptr() % 16 == 0 and ldc % 4 == 0 and (R is None or (R.data_ptr() % 16 == 0 and ldr % 4 == 0)):
        import ctypes

        m0 = M // 128 * 128
        arr = (_lib.GemmWpProblem * 1)()
        q = arr[0]
        q.A, q.lda, q.W, q.plane_stride, q.ldw, q.C, q.ldc = A.data_ptr(), lda, wp[0], wp[1], ldw, C.data_ptr(), ldc
        q.bias, q.R, q.ldr = (None if bias is None else bias.data_ptr()), (None if R is None else R.data_ptr()), (0 if R is None else ldr)
        q.M, q.N, q.K, q.relu = m0, N, K, relu
        _c("rt_gemm_wp", ctypes.cast(arr, ctypes.c_void_p), 1, 0 if w_kc else 1, tag=(m0, N, K), timed_as="rt_gemm")
    if m0 < M:
        _gemm(A[m0:], lda, 1, W, ldw, w_kc, C[m0:], ldc, bias, None if R is None else R[m0:], ldr, M - m0, N, K, relu)


class _Linear(torch.autograd.Function):
    """y = x @ W^T (+ b) (+ residual) (relu).  x [M,K] (row stride free), W [N,K] contiguous."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual, relu):
        _chk(x, "linear")
        M, K = x.shape
        N = weight.shape[0]
        y = torch.empty((M, N), dtype=torch.float32, device=x.device)
        wp = _planes_of(weight) if x.stride(1) == 1 else None
        _gemm_w(x, x.stride(0), weight, weight.stride(0), 1, wp, y, N, bias, residual,
                0 if residual is None else residual.stride(0), M, N, K, 1 if relu else 0)
        ctx.save_for_backward(x, weight, y if relu else None, bias)
        ctx.has_bias, ctx.has_res, ctx.relu, ctx.wp = bias is not None, residual is not None, relu, wp
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, y, bias = ctx.saved_tensors
        M, K = x.shape
        N = weight.shape[0]
        dy = dy.contiguous()
        if ctx.relu:
            dz = torch.empty_like(dy)
            _c("rt_act_dropout_bwd", dy, y, ACT_RELU, 0.0, 0, 0, dy.numel(), dz)
            dy = dz
        dx = dw = db = None
        sd = None
        want_db = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:   # issued first, on the side stream: it overlaps the dgrad product below
            dw = torch.empty((N, K), dtype=torch.float32, device=dy.device)
            if want_db:  # db = colsum(dy) rides on the dy^T tiles of the wgrad product
                db = torch.empty((N,), dtype=torch.float32, device=dy.device)
            sd = _OnSide(dy.device, defer=_steals_grad(weight, bias if want_db else None))
            with sd:
                sd.uses(*(t for t in (dy, x, dw, db) if t is not None))
                _gemm(dy, N, 0, x, x.stride(0), 0, dw, K, None, None, 0, N, K, M, 0, _wgrad_splits(M, N, K), db)  # dW = dy^T @ x
        elif want_db:
            db = torch.zeros((N,), dtype=torch.float32, device=dy.device)
            _c("rt_colsum", dy, N, M, N, db)
        if ctx.needs_input_grad[0]:
            dx = torch.empty((M, K), dtype=torch.float32, device=dy.device)
            _gemm_w(dy, N, weight, weight.stride(0), 0, ctx.wp, dx, K, None, None, 0, M, K, N)  # dx = dy @ W
        if sd is not None:
            sd.join_now()   # parameter slices / accumulating grads: autograd touches dw right after this node
        dres = dy if (ctx.has_res and ctx.needs_input_grad[3]) else None
        return dx, dw, db, dres, None


def linear(x: torch.Tensor, weight: torch.Tensor, bias: tp.Optional[torch.Tensor] = None,
           residual: tp.Optional[torch.Tensor] = None, relu: bool = False) -> torch.Tensor:
    return _Linear.apply(x, weight, bias, residual, relu)


class _MatmulNN(torch.autograd.Function):
    """y = x @ P with P [K,N] contiguous (hstu.py:258, `torch.matmul(normed_x, self.uvqk_proj)`)."""

    @staticmethod
    def forward(ctx, x, p):
        M, K = x.shape
        N = p.shape[1]
        y = torch.empty((M, N), dtype=torch.float32, device=x.device)
        ctx.wp = _planes_of(p) if x.stride(1) == 1 else None
        _gemm_w(x, x.stride(0), p, N, 0, ctx.wp, y, N, None, None, 0, M, N, K)
        ctx.save_for_backward(x, p)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, p = ctx.saved_tensors
        M, K = x.shape
        N = p.shape[1]
        dy = dy.contiguous()
        dp = torch.empty((K, N), dtype=torch.float32, device=dy.device)
        sd = _OnSide(dy.device, defer=_steals_grad(p))
        with sd:
            sd.uses(x, dy, dp)
            _gemm(x, x.stride(0), 0, dy, N, 0, dp, N, None, None, 0, K, N, M, 0, _wgrad_splits(M, K, N))  # dP = x^T @ dy
        dx = torch.empty((M, K), dtype=torch.float32, device=dy.device)
        _gemm_w(dy, N, p, N, 1, ctx.wp, dx, K, None, None, 0, M, K, N)  # dx = dy @ P^T : B(k', n) = P[k'*N + n] (kc)
        sd.join_now()
        return dx, dp


def matmul_nn(x: torch.Tensor, p: torch.Tensor) -> torch.Tensor:
    return _MatmulNN.apply(x, p)


# --------------------------------------------------------------------------------------------------
# row-wise
# --------------------------------------------------------------------------------------------------
# Gradient sink of an item table: the loss node writes d(loss)/d(table) [V,d] first; the embedding node — the LAST node of
# every backward pass — then adds its rows INTO that tensor (rt_embed_bwd accumulate) and returns no gradient of its own,
# instead of producing a second [V,d] tensor that autograd would add with a full-size kernel.  Keyed by the table's storage.
_TABLE_GRAD_SINK: tp.Dict[int, torch.Tensor] = {}
_TABLE_GRAD_ON_SIDE: tp.Set[int] = set()      # sinks whose loss half is in flight on the side stream: the embedding adds there too
# The table half of the sampled losses' backward (counting sort + gathered row reductions: read by the optimiser only) on the side stream,
# the embedding backward behind it.  Round 3 measured it equal (69.3 vs 70.2 k seqs/s at C2: the gathers contend with the layer backward);
# with the row-resident chain kernels (212 workgroups on 256 CUs) and the embedding's sort done ahead of time (RT_PREPARE_AHEAD) it is
# 85.1 vs 81.9 k seqs/s on one box, 84.1 vs 83.2 k on another (round 4).  (The switch RT_LOSS_SIDE was retired in round 6.)
_LOSS_TABLE_ON_SIDE = True


# Tables whose lookup ran in THIS forward pass (weak references, keyed by storage): only then will an embedding node pick the loss's table
# gradient up as its sink and drop the extra reference `_TABLE_GRAD_SINK` holds — without a consumer that reference makes autograd's
# AccumulateGrad CLONE the gradient on the main stream, which must not happen while the side stream is still writing it (seen as a
# rare wrong d_table in a loss-only test once the side stream became the default for the loss's table half).
_TABLE_SINK_EXPECTED: tp.Dict[int, tp.Any] = {}


def _expect_table_sink(table: torch.Tensor) -> None:
    import weakref

    _TABLE_SINK_EXPECTED[table.data_ptr()] = (weakref.ref(table), RNG.step)


def _table_sink_expected(table: torch.Tensor) -> bool:
    """Did an embedding lookup of THIS forward pass register for the table's gradient?  An entry a previous step left behind — its
    loss was a softmax / a plugged one, or its backward never ran — is not an expectation of this step: entries carry the
    step counter of the dropout streams, and `clear_step_expectations()` (FlatAdam.zero_grad) drops what a finished step left."""
    r = _TABLE_SINK_EXPECTED.pop(table.data_ptr(), None)
    if r is None or r[1] != RNG.step:
        return False
    t = r[0]()
    return t is not None and t.data_ptr() == table.data_ptr() and t.shape == table.shape


def clear_step_expectations() -> None:
    """Start of a training step (`FlatAdam.zero_grad`): nothing a previous forward pass registered survives into this one."""
    _TABLE_SINK_EXPECTED.clear()
    _TABLE_GRAD_SINK.clear()
    _TABLE_HOME_CLAIMED.clear()


# Data-parallel runs: where a table's dense gradient should be PRODUCED — its segment of the optimiser's flat gradient buffer
# (`lightning.FlatAdam.gather_gradients` registers it) — so that packing the gradients for the collective does not copy the largest of
# them (1 GB per step at C4).  Keyed by the table's data pointer; the value makes a FRESH view per call (autograd adopts a gradient
# only while nobody else holds the tensor object).
_TABLE_GRAD_HOME: tp.Dict[int, tp.Callable[[], tp.Optional[torch.Tensor]]] = {}


_TABLE_HOME_CLAIMED: tp.Set[int] = set()      # tables whose home view was handed out in THIS step (reset by `clear_step_expectations`)


def _new_table_grad(table: torch.Tensor) -> torch.Tensor:
    """The buffer a loss node writes the table's dense gradient into: the table's home in the flat gradient buffer — at most ONCE per
    step.  A second loss node on the same leaf table in one backward pass (a plugged loss over two augmented views) gets its
    own tensor: two nodes writing the same memory would hand autograd two aliases of it, summed to 2 dB instead of dA + dB."""
    key = table.data_ptr()
    home = _TABLE_GRAD_HOME.get(key)
    if home is not None and key not in _TABLE_HOME_CLAIMED and table.is_leaf and table.grad is None:   # (a second backward pass accumulates: not in place)
        g = home()
        if g is not None and g.shape == table.shape and g.device == table.device and g.dtype == table.dtype:
            _TABLE_HOME_CLAIMED.add(key)
            return g
    return torch.empty_like(table)


def _offer_table_grad(table: torch.Tensor, d_table: torch.Tensor) -> None:
    if d_table.is_contiguous() and d_table.shape == table.shape:
        _TABLE_GRAD_SINK[table.data_ptr()] = d_table


class _Embed(torch.autograd.Function):
    @staticmethod
    def forward(ctx, table, pos, ids, L, scale, p):
        M = ids.numel()
        d = table.shape[1]
        out = torch.empty((M, d), dtype=torch.float32, device=table.device)
        seed, sid = RNG.next() if p > 0 else (0, 0)
        _c("rt_embed_fwd", ids, table, pos, float(scale), M, L, d, float(p), seed, sid, out)
        ctx.save_for_backward(ids)
        ctx.meta = (table.shape, None if pos is None else pos.shape, L, scale, p, seed, sid)
        ctx.table_ptr = table.data_ptr()
        _TABLE_GRAD_SINK.pop(ctx.table_ptr, None)   # a sink left over from an aborted backward pass must not be reused
        if ctx.needs_input_grad[0]:
            _expect_table_sink(table)
        return out

    @staticmethod
    def backward(ctx, gout):
        (ids,) = ctx.saved_tensors
        tshape, pshape, L, scale, p, seed, sid = ctx.meta
        gout = gout.contiguous()
        M, V = ids.numel(), tshape[0]
        sink = _TABLE_GRAD_SINK.pop(ctx.table_ptr, None)
        if sink is not None and (tuple(sink.shape) != tuple(tshape) or sink.device != gout.device):
            sink = None
        gtable = sink if sink is not None else torch.empty(tshape, dtype=torch.float32, device=gout.device)
        gpos = None
        if pshape is not None:  # rows [0, L) are written; a longer table keeps zero gradient behind them
            gpos = (torch.empty if pshape[0] == L else torch.zeros)(pshape, dtype=torch.float32, device=gout.device)
        ws_bytes = _lib.load().rt_embed_bwd_workspace_bytes(M, V, tshape[1])
        ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=gout.device)
        side = None
        if sink is not None and sink.data_ptr() in _TABLE_GRAD_ON_SIDE:   # the sink's loss half runs on the side stream: follow it there
            _TABLE_GRAD_ON_SIDE.discard(sink.data_ptr())
            side = _native_side_fork()
            if side is None:
                join_side_streams()
        _c("rt_embed_bwd", ids, gout, float(scale), M, L, tshape[1], V, float(p), seed, sid, gtable, 1 if sink is not None else 0,
           gpos, ws, ws_bytes, 0, stream=side)
        if side is not None:
            _NATIVE_KEEPALIVE.append((ids, gout, ws))         # not gtable / gpos: autograd must adopt them, not clone them
        return (None if sink is not None else gtable), gpos, None, None, None, None


def embed(table: torch.Tensor, pos: tp.Optional[torch.Tensor], ids: torch.Tensor, L: int, scale: float,
          p: float) -> torch.Tensor:
    """[M,d] = dropout(table[ids] * scale + pos[L-1-l]); ids int64 [M] (or [B,L])."""
    return _Embed.apply(table, pos, ids.reshape(-1), L, scale, p)


class _EmbedPacked(torch.autograd.Function):
    """`_Embed` on packed rows (`rt_embed_packed_fwd / _bwd`): the positional row of a row is pos[dist[row]]."""

    @staticmethod
    def forward(ctx, table, pos, ids, dist, cu, B, L, scale, p):
        M = ids.numel()
        d = table.shape[1]
        out = torch.empty((M, d), dtype=torch.float32, device=table.device)
        seed, sid = RNG.next() if p > 0 else (0, 0)
        _c("rt_embed_packed_fwd", ids, dist, table, pos, float(scale), M, d, float(p), seed, sid, out)
        ws = None
        if ctx.needs_input_grad[0]:
            # the backward's counting sort of the rows by id depends on `ids` alone: issued NOW on the side stream it is long done when
            # the backward pass arrives (eight small launches less in the tail of a step)
            ws_bytes = _lib.load().rt_embed_bwd_workspace_bytes(M, table.shape[0], d)
            ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=table.device)
            # buffers of earlier forward passes whose backward never came (evaluation under grad mode, a plugged loop): join and let
            # them go instead of pinning one workspace per call
            global _PREP_STEP
            if _PREP_KEEPALIVE and (_PREP_STEP != RNG.step or len(_PREP_KEEPALIVE) >= 4):
                join_side_streams()
            side = _side_fork_forward()
            _c("rt_embed_bwd_prepare", ids, M, d, table.shape[0], ws, ws_bytes, stream=side)
            if side is not None:
                _PREP_KEEPALIVE.append((ids, ws))
                _PREP_STEP = RNG.step
        ctx.save_for_backward(ids, cu, *(() if ws is None else (ws,)))
        ctx.meta = (table.shape, None if pos is None else pos.shape, B, L, scale, p, seed, sid)
        ctx.table_ptr = table.data_ptr()
        _TABLE_GRAD_SINK.pop(ctx.table_ptr, None)
        if ctx.needs_input_grad[0]:
            _expect_table_sink(table)
        return out

    @staticmethod
    def backward(ctx, gout):
        ids, cu, *prep = ctx.saved_tensors
        tshape, pshape, B, L, scale, p, seed, sid = ctx.meta
        gout = gout.contiguous()
        M, V = ids.numel(), tshape[0]
        sink = _TABLE_GRAD_SINK.pop(ctx.table_ptr, None)
        if sink is not None and (tuple(sink.shape) != tuple(tshape) or sink.device != gout.device):
            sink = None
        gtable = sink if sink is not None else torch.empty(tshape, dtype=torch.float32, device=gout.device)
        gpos = None
        if pshape is not None:
            gpos = (torch.empty if pshape[0] == L else torch.zeros)(pshape, dtype=torch.float32, device=gout.device)
        ws_bytes = _lib.load().rt_embed_bwd_workspace_bytes(M, V, tshape[1])
        ws = prep[0] if prep else torch.empty((ws_bytes,), dtype=torch.uint8, device=gout.device)
        prep_joined = False
        if sink is not None and sink.data_ptr() in _TABLE_GRAD_ON_SIDE:
            _TABLE_GRAD_ON_SIDE.discard(sink.data_ptr())
            # on the main stream, behind the point the loss's table half (and the rows' counting sort, issued before it) reached on the
            # side stream — not behind the weight gradients that stream was given afterwards (queued on the side stream itself the
            # embedding backward ended 78 us after the main stream: profiles/r4_timeline_train.txt)
            _lib.check(_lib.load().rt_side_wait_mark(_lib.current_stream()), "rt_side_wait_mark")
            prep_joined = True
        if prep and _PREP_KEEPALIVE and not prep_joined:
            join_side_streams()          # the sort ran on the side stream and nothing has joined it yet (no sampled loss in this step)
        _c("rt_embed_packed_bwd", ids, cu, B, gout, float(scale), M, L, tshape[1], V, float(p), seed, sid, gtable,
           1 if sink is not None else 0, gpos, ws, ws_bytes, 1 if prep else 0)
        return (None if sink is not None else gtable), gpos, None, None, None, None, None, None, None


def embed_packed(table: torch.Tensor, pos: tp.Optional[torch.Tensor], ids: torch.Tensor, dist: torch.Tensor, cu: torch.Tensor, B: int,
                 L: int, scale: float, p: float) -> torch.Tensor:
    """[Np, d] = dropout(table[ids] * scale + pos[dist]) over packed rows (ids / dist [Np] from `rt_collate_packed`; cu [B+1])."""
    return _EmbedPacked.apply(table, pos, ids.reshape(-1), dist.reshape(-1), cu, B, L, scale, p)


def collate_packed(offsets: torch.Tensor, items: torch.Tensor, weights: tp.Optional[torch.Tensor], idx: torch.Tensor, cu: torch.Tensor,
                   rows: int, train: bool) -> tp.Tuple[torch.Tensor, ...]:
    """`rt_collate_packed`: -> (x, dist) or, train, (x, y, yw, dist), each [rows] (rows >= cu[-1], the tail zero-filled)."""
    dev = offsets.device
    B = int(idx.numel())
    x = torch.empty((rows,), dtype=torch.int64, device=dev)
    dist = torch.empty((rows,), dtype=torch.int64, device=dev)
    y = torch.empty((rows,), dtype=torch.int64, device=dev) if train else None
    yw = torch.empty((rows,), dtype=torch.float32, device=dev) if train else None
    _c("rt_collate_packed", offsets, items, weights if train else None, idx, cu, B, rows, 1 if train else 0, x, y, yw, dist)
    return (x, y, yw, dist) if train else (x, dist)


def collate_packed_bert(offsets: torch.Tensor, items: torch.Tensor, weights: tp.Optional[torch.Tensor], idx: torch.Tensor, cu: torch.Tensor,
                        rows: int, window: int, train: bool, mask_id: int, probs: tp.Optional[torch.Tensor] = None,
                        rand_ids: tp.Optional[torch.Tensor] = None, mask_prob: float = 0.0,
                        draw_rows: tp.Optional[torch.Tensor] = None) -> tp.Tuple[torch.Tensor, ...]:
    """`rt_collate_packed_bert`: -> (x, dist) or, train, (x, y, yw, dist), each [rows]."""
    dev = offsets.device
    B = int(idx.numel())
    x = torch.empty((rows,), dtype=torch.int64, device=dev)
    dist = torch.empty((rows,), dtype=torch.int64, device=dev)
    y = torch.empty((rows,), dtype=torch.int64, device=dev) if train else None
    yw = torch.empty((rows,), dtype=torch.float32, device=dev) if train else None
    _c("rt_collate_packed_bert", offsets, items, weights if train else None, idx, cu, B, rows, int(window), 1 if train else 0,
       probs, rand_ids, draw_rows, float(mask_prob), int(mask_id), x, y, yw, dist)
    return (x, y, yw, dist) if train else (x, dist)


class BagStructure:
    """Static item -> category-value structure of a CatFeaturesItemNet, plus its transpose cut into chunks for the
    backward reduction (include/rectools_hip.h, K1b).  Built once per model from the reference's three buffers."""

    CHUNK = 128   # entries per wave of the backward's first pass

    def __init__(self, emb_bag_inputs: torch.Tensor, offsets: torch.Tensor, input_lengths: torch.Tensor, n_values: int) -> None:
        import numpy as np

        dev = emb_bag_inputs.device
        inp = emb_bag_inputs.detach().cpu().numpy().astype(np.int64)
        off = offsets.detach().cpu().numpy().astype(np.int64)
        lens = input_lengths.detach().cpu().numpy().astype(np.int64)
        if off.shape != lens.shape or (lens < 0).any() or (off < 0).any() or (len(off) and (off + lens).max() > len(inp)):
            raise ValueError("CatFeaturesItemNet: offsets / input_lengths do not describe slices of emb_bag_inputs")
        # (item, value) pairs in the order the forward reads them
        item_of = np.repeat(np.arange(len(off), dtype=np.int64), lens)
        pos = (np.arange(int(lens.sum()), dtype=np.int64) - np.repeat(np.cumsum(lens) - lens, lens)) + np.repeat(off, lens)
        val_of = inp[pos]
        if len(val_of) and (val_of.min() < 0 or val_of.max() >= n_values):
            raise ValueError("CatFeaturesItemNet: emb_bag_inputs holds ids outside [0, n_cat_feature_values)")
        order = np.lexsort((item_of, val_of))     # grouped by value, ascending item id inside a group
        t_items, t_vals = item_of[order], val_of[order]
        counts = np.bincount(t_vals, minlength=n_values).astype(np.int64)
        starts = np.cumsum(counts) - counts
        n_ch = (counts + self.CHUNK - 1) // self.CHUNK
        feat_chunk_ptr = np.concatenate([[0], np.cumsum(n_ch)]).astype(np.int64)
        ch_val = np.repeat(np.arange(n_values, dtype=np.int64), n_ch)
        ch_idx = np.arange(int(n_ch.sum()), dtype=np.int64) - np.repeat(feat_chunk_ptr[:-1], n_ch)
        ch_start = starts[ch_val] + ch_idx * self.CHUNK
        ch_end = np.minimum(ch_start + self.CHUNK, starts[ch_val] + counts[ch_val])
        chunk_ptr = np.concatenate([ch_start, ch_end[-1:]]) if len(ch_start) else np.zeros(1, np.int64)
        self.n_values, self.n_chunks = n_values, int(len(ch_start))
        self.t_items = torch.from_numpy(np.ascontiguousarray(t_items)).to(dev)
        self.chunk_ptr = torch.from_numpy(np.ascontiguousarray(chunk_ptr.astype(np.int64))).to(dev)
        self.feat_chunk_ptr = torch.from_numpy(feat_chunk_ptr).to(dev)
        self.inputs = emb_bag_inputs.to(torch.int64).contiguous()
        self.offsets = offsets.to(torch.int64).contiguous()
        self.lengths = input_lengths.to(torch.int64).contiguous()


class _ItemTable(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ids_emb, cat_emb, bag, p):
        V, d = bag.offsets.numel(), cat_emb.shape[1]
        if ids_emb is not None and tuple(ids_emb.shape) != (V, d):
            raise ValueError(f"item table: id embeddings {tuple(ids_emb.shape)} do not match the feature structure ({V}, {d})")
        out = torch.empty((V, d), dtype=torch.float32, device=cat_emb.device)
        seed, sid = RNG.next() if p > 0 else (0, 0)
        _c("rt_bag_sum_fwd", ids_emb, cat_emb, bag.inputs, bag.offsets, bag.lengths, V, d, float(p), seed, sid, out)
        ctx.meta = (bag, p, seed, sid, ids_emb is not None, tuple(cat_emb.shape))
        return out

    @staticmethod
    def backward(ctx, dE):
        bag, p, seed, sid, has_ids, cshape = ctx.meta
        dE = dE.contiguous()
        d_cat = torch.empty(cshape, dtype=torch.float32, device=dE.device)    # every row is written by the kernel
        ws_bytes = _lib.load().rt_bag_sum_bwd_workspace_bytes(bag.n_chunks, cshape[1])
        ws = torch.empty((max(ws_bytes, 4),), dtype=torch.uint8, device=dE.device)
        _c("rt_bag_sum_bwd", dE, bag.t_items, bag.chunk_ptr, bag.n_chunks, bag.feat_chunk_ptr, cshape[0], cshape[1], float(p),
           seed, sid, d_cat, ws, ws_bytes)
        return (dE if has_ids else None), d_cat, None, None


def item_table(ids_emb: tp.Optional[torch.Tensor], cat_emb: torch.Tensor, bag: BagStructure, p: float) -> torch.Tensor:
    """[V,d] catalog matrix = ids_emb + dropout(bag sums of cat_emb) (item_net.py:101-132,266-281,463-482)."""
    _chk(cat_emb, "item_table")
    return _ItemTable.apply(ids_emb, cat_emb, bag, p)


def _ln_fwd(x, ids, w, b, eps, M, d, x0, y, mean, rstd, cols):
    """One LayerNorm forward launch.  cols = (grp, grp_real): the rows carry zero columns — column c exists iff c % grp < grp_real
    (`nn.DimPlan`: a model width / head size the kernels cannot tile runs padded) — statistics over the real columns only."""
    if cols is not None:
        _c("rt_layernorm_fwd_cols", x, ids, w, b, float(eps), M, d, int(cols[0]), int(cols[1]), x0, y, mean, rstd)
    elif ids is not None:
        _c("rt_layernorm_fwd_masked", x, ids, w, b, float(eps), M, d, x0, y, mean, rstd)
    else:
        _c("rt_layernorm_fwd", x, w, b, float(eps), M, d, y, mean, rstd)


def _ln_bwd(dy, x, w, mean, rstd, res, ids, mask_dy, mask_dx, M, d, dx, dw, db, cols, bias=None):
    """bias: the LayerNorm's bias parameter — when autograd will merely ADOPT dw / db (`_steals_grad(w, bias)`) their combine pass
    (read by the optimiser only) goes to the library's side stream: one launch and one kernel boundary less between the row kernel
    and the next backward kernel."""
    ws_bytes = _lib.load().rt_layernorm_bwd_workspace_bytes(M, d)
    ws = torch.empty((max(ws_bytes, 4),), dtype=torch.uint8, device=x.device)
    if cols is None and bias is not None and M > 0 and _steals_grad(w, bias):
        _c("rt_layernorm_bwd_rows", dy, x, w, mean, rstd, res, ids, mask_dy, mask_dx, M, d, dx, ws, ws_bytes)
        side = _native_side_fork()            # (None: the side stream is off — the combine follows on this stream)
        _c("rt_layernorm_bwd_combine", ws, ws_bytes, M, d, dw, db, stream=side)
        if side is not None:
            _NATIVE_KEEPALIVE.append(ws)      # NOT dw / db: an extra reference would make AccumulateGrad clone them (on the main stream)
        return
    if cols is not None:
        _c("rt_layernorm_bwd_cols", dy, x, w, mean, rstd, res, ids, mask_dy, mask_dx, M, d, int(cols[0]), int(cols[1]), dx, dw, db, ws, ws_bytes)
    else:
        _c("rt_layernorm_bwd_fused", dy, x, w, mean, rstd, res, ids, mask_dy, mask_dx, M, d, dx, dw, db, ws, ws_bytes)


def _ln_bwd_split(dy, x, w, mean, rstd, res, ids, mask_dy, mask_dx, M, d, dx, dw, db, ws, ws_bytes, on_side: bool) -> None:
    """`rt_layernorm_bwd_fused`, or — on_side: autograd will merely adopt dw / db — its row pass here and its combine pass on the library's
    side stream (`rt_layernorm_bwd_rows` / `_combine`; `ws` is kept alive until the join)."""
    if not on_side or M <= 0:
        _c("rt_layernorm_bwd_fused", dy, x, w, mean, rstd, res, ids, mask_dy, mask_dx, M, d, dx, dw, db, ws, ws_bytes)
        return
    _c("rt_layernorm_bwd_rows", dy, x, w, mean, rstd, res, ids, mask_dy, mask_dx, M, d, dx, ws, ws_bytes)
    side = _native_side_fork()
    _c("rt_layernorm_bwd_combine", ws, ws_bytes, M, d, dw, db, stream=side)
    if side is not None:
        _NATIVE_KEEPALIVE.append(ws)


class _LayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, eps, cols=None):
        x = x.contiguous()
        M, d = x.shape
        y = torch.empty_like(x)
        mean = torch.empty((M,), dtype=torch.float32, device=x.device)
        rstd = torch.empty((M,), dtype=torch.float32, device=x.device)
        _ln_fwd(x, None, w, b, eps, M, d, None, y, mean, rstd, cols)
        ctx.save_for_backward(x, w, mean, rstd)
        ctx.cols, ctx.bias = cols, b
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, mean, rstd = ctx.saved_tensors
        M, d = x.shape
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        dw = torch.empty_like(w)
        db = torch.empty_like(w)
        _ln_bwd(dy, x, w, mean, rstd, None, None, 0, 0, M, d, dx, dw, db, ctx.cols, bias=ctx.bias)
        return dx, dw, db, None, None


class _LayerNormMasked(torch.autograd.Function):
    """LN(x * (ids != 0)) — the timeline mask in front of the stack's last LayerNorm (sasrec.py:313-314) applied inside the
    LayerNorm kernels: forward reads x once, backward writes the masked dx directly."""

    @staticmethod
    def forward(ctx, x, ids, w, b, eps, cols=None):
        x = x.contiguous()
        M, d = x.shape
        x0, y = torch.empty_like(x), torch.empty_like(x)
        mean = torch.empty((M,), dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        _ln_fwd(x, ids, w, b, eps, M, d, x0, y, mean, rstd, cols)
        ctx.save_for_backward(x0, ids, w, mean, rstd)
        ctx.cols, ctx.bias = cols, b
        return y

    @staticmethod
    def backward(ctx, dy):
        x0, ids, w, mean, rstd = ctx.saved_tensors
        M, d = x0.shape
        dy = dy.contiguous()
        dx, dw, db = torch.empty_like(x0), torch.empty_like(w), torch.empty_like(w)
        _ln_bwd(dy, x0, w, mean, rstd, None, ids, 0, 1, M, d, dx, dw, db, ctx.cols, bias=ctx.bias)
        return dx, None, dw, db, None, None


def layer_norm_masked(x: torch.Tensor, ids: torch.Tensor, w: torch.Tensor, b: torch.Tensor, eps: float,
                      cols: tp.Optional[tp.Tuple[int, int]] = None) -> torch.Tensor:
    return _LayerNormMasked.apply(_chk(x, "layer_norm_masked"), ids.reshape(-1), w, b, eps, cols)


class _LayerNormSkip(torch.autograd.Function):
    """(LN(x), x): the second output is the skip branch of a pre-LN block.  Its gradient is added inside the LayerNorm backward
    kernel (`rt_layernorm_bwd_fused`, res) instead of by a full-size add kernel that autograd issues for a tensor with two
    consumers."""

    @staticmethod
    def forward(ctx, x, w, b, eps, cols=None):
        x = x.contiguous()
        M, d = x.shape
        y = torch.empty_like(x)
        mean = torch.empty((M,), dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        _ln_fwd(x, None, w, b, eps, M, d, None, y, mean, rstd, cols)
        ctx.save_for_backward(x, w, mean, rstd)
        ctx.cols = cols
        return y, x.view_as(x)

    @staticmethod
    def backward(ctx, dy, dskip):
        x, w, mean, rstd = ctx.saved_tensors
        M, d = x.shape
        if dy is None:
            return dskip, None, None, None, None
        dy = dy.contiguous()
        res = None if dskip is None else dskip.contiguous()
        dx, dw, db = torch.empty_like(x), torch.empty_like(w), torch.empty_like(w)
        _ln_bwd(dy, x, w, mean, rstd, res, None, 0, 0, M, d, dx, dw, db, ctx.cols)
        return dx, dw, db, None, None


def layer_norm_skip(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, eps: float,
                    cols: tp.Optional[tp.Tuple[int, int]] = None) -> tp.Tuple[torch.Tensor, torch.Tensor]:
    """-> (LayerNorm(x), x'): use x' for the skip connection / every other consumer of x."""
    return _LayerNormSkip.apply(_chk(x, "layer_norm_skip"), w, b, eps, cols)


class _DropoutAdd(torch.autograd.Function):
    """residual + dropout(x) in one pass (net_blocks.py:257-259)."""

    @staticmethod
    def forward(ctx, x, residual, p):
        x = x.contiguous()
        y = torch.empty_like(x)
        seed = RNG.next() if p > 0 else (0, 0)
        _c("rt_act_dropout_fwd", x, ACT_NONE, float(p), seed[0], seed[1], x.numel(), residual.contiguous(), y)
        ctx.meta = (p, seed)
        return y

    @staticmethod
    def backward(ctx, dy):
        p, seed = ctx.meta
        dy = dy.contiguous()
        if p > 0:
            dx = torch.empty_like(dy)
            _c("rt_act_dropout_bwd", dy, dy, ACT_NONE, float(p), seed[0], seed[1], dy.numel(), dx)
        else:
            dx = dy
        return dx, dy, None


def dropout_add(x: torch.Tensor, residual: torch.Tensor, p: float) -> torch.Tensor:
    return _DropoutAdd.apply(_chk(x, "dropout_add"), residual, p)


def layer_norm(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, eps: float, cols: tp.Optional[tp.Tuple[int, int]] = None) -> torch.Tensor:
    return _LayerNorm.apply(x, w, b, eps, cols)


class _ActDropout(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, kind, p):
        z = z.contiguous()
        y = torch.empty_like(z)
        seed, sid = RNG.next() if p > 0 else (0, 0)
        _c("rt_act_dropout_fwd", z, kind, float(p), seed, sid, z.numel(), None, y)
        ctx.save_for_backward(z)
        ctx.meta = (kind, p, seed, sid)
        return y

    @staticmethod
    def backward(ctx, dy):
        (z,) = ctx.saved_tensors
        kind, p, seed, sid = ctx.meta
        dy = dy.contiguous()
        dz = torch.empty_like(z)
        _c("rt_act_dropout_bwd", dy, z, kind, float(p), seed, sid, z.numel(), dz)
        return dz, None, None


def act_dropout(z: torch.Tensor, kind: int, p: float) -> torch.Tensor:
    if kind == ACT_NONE and p <= 0:
        return z
    return _ActDropout.apply(z, kind, p)


def dropout(x: torch.Tensor, p: float) -> torch.Tensor:
    return act_dropout(x, ACT_NONE, p)


class _Swiglu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, p):
        a, b = a.contiguous(), b.contiguous()
        y = torch.empty_like(a)
        seed, sid = RNG.next() if p > 0 else (0, 0)
        _c("rt_swiglu_fwd", a, b, float(p), seed, sid, a.numel(), y)
        ctx.save_for_backward(a, b)
        ctx.meta = (p, seed, sid)
        return y

    @staticmethod
    def backward(ctx, dy):
        a, b = ctx.saved_tensors
        p, seed, sid = ctx.meta
        dy = dy.contiguous()
        da, db = torch.empty_like(a), torch.empty_like(b)
        _c("rt_swiglu_bwd", dy, a, b, float(p), seed, sid, a.numel(), da, db)
        return da, db, None


def swiglu(a: torch.Tensor, b: torch.Tensor, p: float) -> torch.Tensor:
    return _Swiglu.apply(a, b, p)


class _Gate(torch.autograd.Function):
    """y = x + sigmoid(gz) * dropout(a)"""

    @staticmethod
    def forward(ctx, x, gz, a, p):
        x, gz, a = x.contiguous(), gz.contiguous(), a.contiguous()
        y = torch.empty_like(x)
        seed, sid = RNG.next() if p > 0 else (0, 0)
        _c("rt_gate_fwd", x, gz, a, float(p), seed, sid, x.numel(), y)
        ctx.save_for_backward(gz, a)
        ctx.meta = (p, seed, sid)
        return y

    @staticmethod
    def backward(ctx, dy):
        gz, a = ctx.saved_tensors
        p, seed, sid = ctx.meta
        dy = dy.contiguous()
        dgz, da = torch.empty_like(gz), torch.empty_like(a)
        _c("rt_gate_bwd", dy, gz, a, float(p), seed, sid, gz.numel(), dgz, da)
        return dy, dgz, da, None


def gate(x: torch.Tensor, gz: torch.Tensor, a: torch.Tensor, p: float) -> torch.Tensor:
    return _Gate.apply(x, gz, a, p)


class _GatedResidual(torch.autograd.Function):
    """y = x + sigmoid(x Wg^T + bg) * dropout(a)   (ligr.py:99-105): the gating linear and the gate as ONE node.  x has a single
    consumer then, and its two gradient contributions (the skip connection and the gating linear's data gradient) meet in the
    residual epilogue of the dgrad GEMM instead of in a full-size add kernel issued by autograd."""

    @staticmethod
    def forward(ctx, x, wg, bg, a, p):
        x, a = x.contiguous(), a.contiguous()
        M, d = x.shape
        gz = torch.empty_like(x)
        _gemm(x, d, 1, wg, d, 1, gz, d, bg, None, 0, M, d, d)
        y = torch.empty_like(x)
        seed, sid = RNG.next() if p > 0 else (0, 0)
        _c("rt_gate_fwd", x, gz, a, float(p), seed, sid, x.numel(), y)
        ctx.save_for_backward(x, wg, gz, a)
        ctx.meta = (p, seed, sid, bg is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, wg, gz, a = ctx.saved_tensors
        p, seed, sid, has_bias = ctx.meta
        dy = dy.contiguous()
        M, d = x.shape
        dgz, da = torch.empty_like(gz), torch.empty_like(a)
        _c("rt_gate_bwd", dy, gz, a, float(p), seed, sid, gz.numel(), dgz, da)
        dw = torch.empty_like(wg)
        db = torch.empty((d,), dtype=torch.float32, device=dy.device) if has_bias else None
        sd = _OnSide(dy.device, defer=_steals_grad(wg))
        with sd:
            sd.uses(*(t for t in (dgz, x, dw, db) if t is not None))
            _gemm(dgz, d, 0, x, d, 0, dw, d, None, None, 0, d, d, M, 0, _wgrad_splits(M, d, d), db)
        dx = torch.empty_like(x)
        _gemm(dgz, d, 1, wg, d, 0, dx, d, None, dy, d, M, d, d)       # dgz Wg + dy (the skip connection)
        sd.join_now()
        return dx, dw, db, da, None


def gated_residual(x: torch.Tensor, wg: torch.Tensor, bg: tp.Optional[torch.Tensor], a: torch.Tensor, p: float) -> torch.Tensor:
    return _GatedResidual.apply(_chk(x, "gated_residual"), wg, bg, a, p)


class _MulMask(torch.autograd.Function):
    """y = a * b * (ids != 0); b / ids optional"""

    @staticmethod
    def forward(ctx, a, b, ids):
        a = a.contiguous()
        b = None if b is None else b.contiguous()
        y = torch.empty_like(a)
        _c("rt_mul_mask", a, b, ids, a.shape[1], a.numel(), y)
        ctx.save_for_backward(a if b is not None else None, b, ids)
        return y

    @staticmethod
    def backward(ctx, dy):
        a, b, ids = ctx.saved_tensors
        dy = dy.contiguous()
        da = torch.empty_like(dy)
        _c("rt_mul_mask", dy, b, ids, dy.shape[1], dy.numel(), da)
        db = None
        if b is not None:
            db = torch.empty_like(dy)
            _c("rt_mul_mask", dy, a, ids, dy.shape[1], dy.numel(), db)
        return da, db, None


def mul_mask(a: torch.Tensor, b: tp.Optional[torch.Tensor], ids: tp.Optional[torch.Tensor]) -> torch.Tensor:
    return _MulMask.apply(a, b, None if ids is None else ids.reshape(-1))


class _Add(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = a.contiguous(), b.contiguous()
        y = torch.empty_like(a)
        _c("rt_axpy", a, 1.0, b, a.numel(), y)
        return y

    @staticmethod
    def backward(ctx, dy):
        return dy, dy


def add(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    return _Add.apply(a, b)


class _L2Norm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        M, d = x.shape
        y = torch.empty((M, d), dtype=torch.float32, device=x.device)
        _c("rt_l2norm_fwd", x, x.stride(0), M, d, y, d, None)
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        M, d = x.shape
        dy = dy.contiguous()
        dx = torch.empty((M, d), dtype=torch.float32, device=dy.device)
        _c("rt_l2norm_bwd", dy, d, x, x.stride(0), M, d, 0, dx, d)
        return dx


def l2norm(x: torch.Tensor) -> torch.Tensor:
    return _L2Norm.apply(x)


# --------------------------------------------------------------------------------------------------
# attention
# --------------------------------------------------------------------------------------------------
class _MHA(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, ids, B, H, L, causal, keypad, p, scale=0.0):
        d = q.shape[1]
        hd = d // H
        o = torch.empty((B * L, d), dtype=torch.float32, device=q.device)
        lse = torch.empty((B, H, L), dtype=torch.float32, device=q.device)
        seed = 0
        if p > 0:
            s0, sid = RNG.next()
            seed = (s0 + 0xD1B54A32D192ED03 * sid) & 0xFFFFFFFFFFFFFFFF
        _c("rt_mha_fwd_scaled", q, q.stride(0), k, k.stride(0), v, v.stride(0), ids, B, H, L, hd, float(scale), int(causal), int(keypad),
           float(p), seed, o, d, lse)
        ctx.save_for_backward(q, k, v, o, lse, ids)
        ctx.meta = (B, H, L, hd, causal, keypad, p, seed, scale)
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, o, lse, ids = ctx.saved_tensors
        B, H, L, hd, causal, keypad, p, seed, scale = ctx.meta
        d = H * hd
        do = do.contiguous()
        dqkv = torch.empty((3, B * L, d), dtype=torch.float32, device=do.device)
        delta = torch.empty((B, H, L), dtype=torch.float32, device=do.device)
        _c("rt_mha_bwd_scaled", q, q.stride(0), k, k.stride(0), v, v.stride(0), o, d, do, d, lse, ids, B, H, L, hd, float(scale), int(causal),
           int(keypad), float(p), seed, dqkv[0], d, dqkv[1], d, dqkv[2], d, delta)
        return dqkv[0], dqkv[1], dqkv[2], None, None, None, None, None, None, None, None


def mha(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, ids: torch.Tensor, B: int, H: int, L: int, causal: bool,
        keypad: bool, p: float, scale: float = 0.0) -> torch.Tensor:
    """Softmax attention over [B*L, d] projections (column slices of packed buffers are fine).  scale: the logit scale (0 = 1 / sqrt(head
    size): `torch.nn.MultiheadAttention`'s; a head padded with zero columns passes the scale of its real size)."""
    return _MHA.apply(q, k, v, ids.reshape(-1), B, H, L, causal, keypad, p, scale)


class _MHAPacked(torch.autograd.Function):
    """Self-attention on a packed in_proj output qkv [M, 3d] (net_blocks.py:248-255, ligr.py:91-98).  The backward WRITES dq, dk,
    dv into the column slices of one [M, 3d] gradient; with three sliced views of qkv, autograd's SliceBackward builds three
    zero-filled [M, 3d] buffers, copies a slice into each and adds them (5 % of the BERT4Rec step in at:: kernels)."""

    @staticmethod
    def forward(ctx, qkv, ids, B, H, L, causal, keypad, p, scale=0.0):
        M, d3 = qkv.shape
        d = d3 // 3
        hd = d // H
        o = torch.empty((M, d), dtype=torch.float32, device=qkv.device)
        lse = torch.empty((B, H, L), dtype=torch.float32, device=qkv.device)
        seed = 0
        if p > 0:
            s0, sid = RNG.next()
            seed = (s0 + 0xD1B54A32D192ED03 * sid) & 0xFFFFFFFFFFFFFFFF
        _c("rt_mha_fwd_scaled", qkv, d3, qkv[:, d:], d3, qkv[:, 2 * d:], d3, ids, B, H, L, hd, float(scale), int(causal), int(keypad), float(p),
           seed, o, d, lse)
        ctx.save_for_backward(qkv, o, lse, ids)
        ctx.meta = (B, H, L, hd, causal, keypad, p, seed, scale)
        return o

    @staticmethod
    def backward(ctx, do):
        qkv, o, lse, ids = ctx.saved_tensors
        B, H, L, hd, causal, keypad, p, seed, scale = ctx.meta
        d = H * hd
        d3 = 3 * d
        do = do.contiguous()
        g = torch.empty_like(qkv)
        delta = torch.empty((B, H, L), dtype=torch.float32, device=do.device)
        _c("rt_mha_bwd_scaled", qkv, d3, qkv[:, d:], d3, qkv[:, 2 * d:], d3, o, d, do, d, lse, ids, B, H, L, hd, float(scale), int(causal),
           int(keypad), float(p), seed, g, d3, g[:, d:], d3, g[:, 2 * d:], d3, delta)
        return g, None, None, None, None, None, None, None, None


def mha_packed(qkv: torch.Tensor, ids: torch.Tensor, B: int, H: int, L: int, causal: bool, keypad: bool, p: float,
               scale: float = 0.0) -> torch.Tensor:
    """Softmax self-attention over a packed [B*L, 3d] projection (q | k | v column blocks); scale as `mha`."""
    return _MHAPacked.apply(_chk(qkv, "mha_packed").contiguous(), ids.reshape(-1), B, H, L, causal, keypad, p, scale)


class _SASRecLayer(torch.autograd.Function):
    """One SASRec block (sasrec.py:186-231, :300) as a single autograd node.

        x0 = x * (ids != 0);  q = LN1(x0);  y = q + out_proj(mha(Wq q, Wkv x0));  f = LN2(y)
        out = f + dropout(W2 dropout(relu(W1 f + b1)) + b2)

    Same kernels as the modular path (ops.linear / layer_norm / mha / ...); what the fused node adds is control over the
    backward data flow: every gradient accumulation (the two residuals and the two consumers of x0) rides in the
    residual epilogue of a dgrad GEMM instead of a separate full-size add kernel issued by autograd.
    """

    @staticmethod
    def forward(ctx, x, ids, ln1_w, ln1_b, in_w, in_b, out_w, out_b, ln2_w, ln2_b, w1, b1, w2, b2, meta):
        B, L, H, causal, keypad, p, eps1, eps2 = meta
        x = x.contiguous()
        M, d = x.shape
        dff = w1.shape[0]
        dev = x.device
        new = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)  # noqa: E731
        x0, q, mean1, rstd1 = new(M, d), new(M, d), new(M), new(M)
        _c("rt_layernorm_fwd_masked", x, ids, ln1_w, ln1_b, float(eps1), M, d, x0, q, mean1, rstd1)   # x0 = x * mask, q = LN1(x0)
        Q, KV = new(M, d), new(M, 2 * d)
        _gemm_group([(q, d, in_w, d, Q, d, in_b, None, 0, M, d, d, 0),                       # Q = LN1(x0) Wq^T + bq
                     (x0, d, in_w[d:], d, KV, 2 * d, in_b[d:], None, 0, M, 2 * d, d, 0)], 1, 1)  # K,V = x0 Wkv^T + bkv
        A, lse = new(M, d), new(B, H, L)
        seed_a = 0
        if p > 0:
            s0, sid = RNG.next()
            seed_a = (s0 + 0xD1B54A32D192ED03 * sid) & 0xFFFFFFFFFFFFFFFF
        hd = d // H
        _c("rt_mha_fwd", Q, d, KV, 2 * d, KV[:, d:], 2 * d, ids, B, H, L, hd, int(causal), int(keypad), float(p), seed_a, A, d, lse)
        y = new(M, d)
        _gemm(A, d, 1, out_w, d, 1, y, d, out_b, q, d, M, d, d)
        f, mean2, rstd2 = new(M, d), new(M), new(M)
        _c("rt_layernorm_fwd", y, ln2_w, ln2_b, float(eps2), M, d, f, mean2, rstd2)
        h = new(M, dff)
        _gemm(f, d, 1, w1, d, 1, h, dff, b1, None, 0, M, dff, d, 1)
        seed_h = seed_o = (0, 0)
        if p > 0:
            seed_h = RNG.next()
            hdrop = new(M, dff)
            _c("rt_act_dropout_fwd", h, ACT_NONE, float(p), seed_h[0], seed_h[1], h.numel(), None, hdrop)
            o = new(M, d)
            _gemm(hdrop, dff, 1, w2, dff, 1, o, d, b2, None, 0, M, d, dff)
            seed_o = RNG.next()
            out = new(M, d)   # dropout and the skip connection in one pass
            _c("rt_act_dropout_fwd", o, ACT_NONE, float(p), seed_o[0], seed_o[1], o.numel(), f, out)
        else:
            hdrop = h
            out = new(M, d)
            _gemm(h, dff, 1, w2, dff, 1, out, d, b2, f, d, M, d, dff)
        ctx.save_for_backward(ids, x0, q, Q, KV, A, lse, y, f, h, hdrop, mean1, rstd1, mean2, rstd2,
                              ln1_w, in_w, out_w, ln2_w, w1, w2)
        ctx.meta = (B, L, H, causal, keypad, p, seed_a, seed_h, seed_o)
        return out

    @staticmethod
    def backward(ctx, g_out):
        (ids, x0, q, Q, KV, A, lse, y, f, h, hdrop, mean1, rstd1, mean2, rstd2, ln1_w, in_w, out_w, ln2_w, w1, w2) = ctx.saved_tensors
        B, L, H, causal, keypad, p, seed_a, seed_h, seed_o = ctx.meta
        g_out = g_out.contiguous()
        M, d = g_out.shape
        dff = w1.shape[0]
        dev = g_out.device
        new = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)  # noqa: E731
        sp = _wgrad_splits(M)
        # weight gradients stay in flight on the side stream only if autograd merely adopts them (see _steals_grad);
        # saved 1-D parameters are not kept, so the check covers the matrices — biases / LN vectors of one block are
        # leaves exactly when its matrices are
        defer = _steals_grad(ln1_w, in_w, out_w, ln2_w, w1, w2)
        side_ctxs = []

        def ln_bwd(dy, x, w, mean, rstd):
            dx, dw, db = new(M, d), new(d), new(d)
            ws_bytes = _lib.load().rt_layernorm_bwd_workspace_bytes(M, d)
            ws = torch.empty((max(ws_bytes, 4),), dtype=torch.uint8, device=dev)
            _c("rt_layernorm_bwd", dy, x, w, mean, rstd, M, d, dx, dw, db, ws, ws_bytes)
            return dx, dw, db

        # ---- feed-forward: out = f + dropout(o),  o = W2 hdrop + b2,  hdrop = dropout(relu(W1 f + b1))
        if p > 0:
            g_o = new(M, d)
            _c("rt_act_dropout_bwd", g_out, g_out, ACT_NONE, float(p), seed_o[0], seed_o[1], g_out.numel(), g_o)
        else:
            g_o = g_out
        d_w2, d_b2 = new(d, dff), new(d)
        with _OnSide(dev, defer) as sd:   # weight gradients leave the critical path (see _OnSide)
            side_ctxs.append(sd)
            sd.uses(g_o, hdrop, d_w2, d_b2)
            _gemm(g_o, d, 0, hdrop, dff, 0, d_w2, dff, None, None, 0, d, dff, M, 0, sp, d_b2)
        g_hd = new(M, dff)
        _gemm(g_o, d, 1, w2, dff, 0, g_hd, dff, None, None, 0, M, dff, d)
        g_h = new(M, dff)   # dropout mask and relu'(h) in one pass (relu'(z) == [h > 0])
        _c("rt_act_dropout_bwd", g_hd, h, ACT_RELU, float(p), seed_h[0], seed_h[1], g_hd.numel(), g_h)
        d_w1, d_b1 = new(dff, d), new(dff)
        with _OnSide(dev, defer) as sd:
            side_ctxs.append(sd)
            sd.uses(g_h, f, d_w1, d_b1)
            _gemm(g_h, dff, 0, f, d, 0, d_w1, d, None, None, 0, dff, d, M, 0, sp, d_b1)
        g_f = new(M, d)     # residual branch (g_out) added in the dgrad epilogue
        _gemm(g_h, dff, 1, w1, d, 0, g_f, d, None, g_out, d, M, d, dff)
        g_y, d_ln2w, d_ln2b = ln_bwd(g_f, y, ln2_w, mean2, rstd2)
        # ---- attention: y = q + Wo A + bo
        d_wo, d_bo = new(d, d), new(d)
        with _OnSide(dev, defer) as sd:
            side_ctxs.append(sd)
            sd.uses(g_y, A, d_wo, d_bo)
            _gemm(g_y, d, 0, A, d, 0, d_wo, d, None, None, 0, d, d, M, 0, sp, d_bo)
        g_A = new(M, d)
        _gemm(g_y, d, 1, out_w, d, 0, g_A, d, None, None, 0, M, d, d)
        gQ, gKV, delta = new(M, d), new(M, 2 * d), new(B, H, L)
        _c("rt_mha_bwd", Q, d, KV, 2 * d, KV[:, d:], 2 * d, A, d, g_A, d, lse, ids, B, H, L, d // H, int(causal), int(keypad),
           float(p), seed_a, gQ, d, gKV, 2 * d, gKV[:, d:], 2 * d, delta)
        d_in_w, d_in_b = new(3 * d, d), new(3 * d)
        with _OnSide(dev, defer) as sd:
            side_ctxs.append(sd)
            sd.uses(gQ, gKV, q, x0, d_in_w, d_in_b)
            _gemm(gQ, d, 0, q, d, 0, d_in_w, d, None, None, 0, d, d, M, 0, sp, d_in_b)
            _gemm(gKV, 2 * d, 0, x0, d, 0, d_in_w[d:], d, None, None, 0, 2 * d, d, M, 0, sp, d_in_b[d:])
        # q feeds the query projection and the residual: g_q = gQ Wq + g_y; x0 feeds LN1 and the key/value projection:
        # g_x = mask * (gKV Wkv + LN1'(g_q)) — the sum and the mask ride in the LayerNorm backward kernel.  One launch for both.
        g_q, g_kv = new(M, d), new(M, d)
        _gemm_group([(gQ, d, in_w, d, g_q, d, None, g_y, d, M, d, d, 0),
                     (gKV, 2 * d, in_w[d:], d, g_kv, d, None, None, 0, M, d, 2 * d, 0)], 1, 0)
        g_x, d_ln1w, d_ln1b = new(M, d), new(d), new(d)
        ws_bytes = _lib.load().rt_layernorm_bwd_workspace_bytes(M, d)
        ws = torch.empty((max(ws_bytes, 4),), dtype=torch.uint8, device=dev)
        _c("rt_layernorm_bwd_fused", g_q, x0, ln1_w, mean1, rstd1, g_kv, ids, 0, 1, M, d, g_x, d_ln1w, d_ln1b, ws, ws_bytes)
        if side_ctxs:
            side_ctxs[-1].join_now()   # no-op when deferred; one event covers every product queued on the side stream
        return (g_x, None, d_ln1w, d_ln1b, d_in_w, d_in_b, d_wo, d_bo, d_ln2w, d_ln2b, d_w1, d_b1, d_w2, d_b2, None)


def sasrec_layer(x: torch.Tensor, ids: torch.Tensor, B: int, L: int, H: int, causal: bool, keypad: bool, p: float,
                 ln1: tp.Tuple[torch.Tensor, torch.Tensor, float], in_proj: tp.Tuple[torch.Tensor, torch.Tensor],
                 out_proj: tp.Tuple[torch.Tensor, torch.Tensor], ln2: tp.Tuple[torch.Tensor, torch.Tensor, float],
                 ff1: tp.Tuple[torch.Tensor, torch.Tensor], ff2: tp.Tuple[torch.Tensor, torch.Tensor]) -> torch.Tensor:
    """Fused SASRec block on [B*L, d] activations (timeline mask included); parameters as (weight, bias[, eps])."""
    for t in (x, ln1[0], in_proj[0], out_proj[0], ln2[0], ff1[0], ff2[0]):
        _chk(t, "sasrec_layer")
    return _SASRecLayer.apply(x, ids.reshape(-1), ln1[0], ln1[1], in_proj[0], in_proj[1], out_proj[0], out_proj[1],
                              ln2[0], ln2[1], ff1[0], ff1[1], ff2[0], ff2[1],
                              (B, L, H, bool(causal), bool(keypad), float(p), ln1[2], ln2[2]))


class _SASRecLayerPacked(torch.autograd.Function):
    """`_SASRecLayer` on PACKED rows: x [Np, d] holds the real positions of B sessions (rows cu[b] .. cu[b+1]-1) and
    an unused tail up to the 128-row tile.  No masks anywhere — there are no pad rows; the pad keys the reference's window shows
    to every query are the virtual key of `rt_mha_varlen_*` (`pad_keys`).  Same products, epilogue fusions and side-stream
    weight gradients as the padded node; the bias gradient of the key / value projection takes the pad keys' share (zero in
    total for b_k, the partials of `rt_mha_varlen_bwd` for b_v)."""

    @staticmethod
    def forward(ctx, x, cu, ln1_w, ln1_b, in_w, in_b, out_w, out_b, ln2_w, ln2_b, w1, b1, w2, b2, meta):
        B, H, window, pad_keys, p, eps1, eps2 = meta
        x = x.contiguous()
        M, d = x.shape
        dff = w1.shape[0]
        dev = x.device
        new = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)  # noqa: E731
        q, mean1, rstd1 = new(M, d), new(M), new(M)
        _c("rt_layernorm_fwd", x, ln1_w, ln1_b, float(eps1), M, d, q, mean1, rstd1)
        Q, KV = new(M, d), new(M, 2 * d)
        _gemm_group([(q, d, in_w, d, Q, d, in_b, None, 0, M, d, d, 0),
                     (x, d, in_w[d:], d, KV, 2 * d, in_b[d:], None, 0, M, 2 * d, d, 0)], 1, 1)
        A = torch.zeros((M, d), dtype=torch.float32, device=dev)        # the kernel writes the sessions' rows only
        lse = torch.zeros((M, H), dtype=torch.float32, device=dev)
        seed_a = 0
        if p > 0:
            s0, sid = RNG.next()
            seed_a = (s0 + 0xD1B54A32D192ED03 * sid) & 0xFFFFFFFFFFFFFFFF
        hd = d // H
        bk, bv = (in_b[d:2 * d], in_b[2 * d:]) if pad_keys else (None, None)
        _c("rt_mha_varlen_train_fwd", Q, d, KV, 2 * d, KV[:, d:], 2 * d, cu, bk, bv, B, H, hd, window, window, float(p), seed_a, A, d, lse)
        y = new(M, d)
        _gemm(A, d, 1, out_w, d, 1, y, d, out_b, q, d, M, d, d)
        f, mean2, rstd2 = new(M, d), new(M), new(M)
        _c("rt_layernorm_fwd", y, ln2_w, ln2_b, float(eps2), M, d, f, mean2, rstd2)
        h = new(M, dff)
        _gemm(f, d, 1, w1, d, 1, h, dff, b1, None, 0, M, dff, d, 1)
        seed_h = seed_o = (0, 0)
        if p > 0:
            seed_h = RNG.next()
            hdrop = new(M, dff)
            _c("rt_act_dropout_fwd", h, ACT_NONE, float(p), seed_h[0], seed_h[1], h.numel(), None, hdrop)
            o = new(M, d)
            _gemm(hdrop, dff, 1, w2, dff, 1, o, d, b2, None, 0, M, d, dff)
            seed_o = RNG.next()
            out = new(M, d)
            _c("rt_act_dropout_fwd", o, ACT_NONE, float(p), seed_o[0], seed_o[1], o.numel(), f, out)
        else:
            hdrop = h
            out = new(M, d)
            _gemm(h, dff, 1, w2, dff, 1, out, d, b2, f, d, M, d, dff)
        ctx.save_for_backward(cu, x, q, Q, KV, A, lse, y, f, h, hdrop, mean1, rstd1, mean2, rstd2,
                              ln1_w, in_w, in_b, out_w, ln2_w, w1, w2)
        ctx.meta = (B, H, window, pad_keys, p, seed_a, seed_h, seed_o)
        return out

    @staticmethod
    def backward(ctx, g_out):
        (cu, x, q, Q, KV, A, lse, y, f, h, hdrop, mean1, rstd1, mean2, rstd2, ln1_w, in_w, in_b, out_w, ln2_w, w1, w2) = ctx.saved_tensors
        B, H, window, pad_keys, p, seed_a, seed_h, seed_o = ctx.meta
        g_out = g_out.contiguous()
        M, d = g_out.shape
        dff = w1.shape[0]
        dev = g_out.device
        new = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)  # noqa: E731
        sp = _wgrad_splits(M)
        defer = _steals_grad(ln1_w, in_w, out_w, ln2_w, w1, w2)
        side_ctxs = []

        def ln_bwd(dy, xx, w, mean, rstd, res=None):
            dx, dw, db = new(M, d), new(d), new(d)
            ws_bytes = _lib.load().rt_layernorm_bwd_workspace_bytes(M, d)
            ws = torch.empty((max(ws_bytes, 4),), dtype=torch.uint8, device=dev)
            _c("rt_layernorm_bwd_fused", dy, xx, w, mean, rstd, res, None, 0, 0, M, d, dx, dw, db, ws, ws_bytes)
            return dx, dw, db

        # ---- feed-forward
        if p > 0:
            g_o = new(M, d)
            _c("rt_act_dropout_bwd", g_out, g_out, ACT_NONE, float(p), seed_o[0], seed_o[1], g_out.numel(), g_o)
        else:
            g_o = g_out
        d_w2, d_b2 = new(d, dff), new(d)
        with _OnSide(dev, defer) as sd:
            side_ctxs.append(sd)
            sd.uses(g_o, hdrop, d_w2, d_b2)
            _gemm(g_o, d, 0, hdrop, dff, 0, d_w2, dff, None, None, 0, d, dff, M, 0, sp, d_b2)
        g_hd = new(M, dff)
        _gemm(g_o, d, 1, w2, dff, 0, g_hd, dff, None, None, 0, M, dff, d)
        g_h = new(M, dff)
        _c("rt_act_dropout_bwd", g_hd, h, ACT_RELU, float(p), seed_h[0], seed_h[1], g_hd.numel(), g_h)
        d_w1, d_b1 = new(dff, d), new(dff)
        with _OnSide(dev, defer) as sd:
            side_ctxs.append(sd)
            sd.uses(g_h, f, d_w1, d_b1)
            _gemm(g_h, dff, 0, f, d, 0, d_w1, d, None, None, 0, dff, d, M, 0, sp, d_b1)
        g_f = new(M, d)
        _gemm(g_h, dff, 1, w1, d, 0, g_f, d, None, g_out, d, M, d, dff)
        g_y, d_ln2w, d_ln2b = ln_bwd(g_f, y, ln2_w, mean2, rstd2)
        # ---- attention: y = q + Wo A + bo
        d_wo, d_bo = new(d, d), new(d)
        with _OnSide(dev, defer) as sd:
            side_ctxs.append(sd)
            sd.uses(g_y, A, d_wo, d_bo)
            _gemm(g_y, d, 0, A, d, 0, d_wo, d, None, None, 0, d, d, M, 0, sp, d_bo)
        g_A = new(M, d)
        _gemm(g_y, d, 1, out_w, d, 0, g_A, d, None, None, 0, M, d, d)
        gQ = torch.zeros((M, d), dtype=torch.float32, device=dev)      # rows behind the sessions must read as zero in the wgrads
        gKV = torch.zeros((M, 2 * d), dtype=torch.float32, device=dev)
        delta = new(M, H)
        part = new(B, d) if pad_keys else None
        bk, bv = (in_b[d:2 * d], in_b[2 * d:]) if pad_keys else (None, None)
        _c("rt_mha_varlen_bwd", Q, d, KV, 2 * d, KV[:, d:], 2 * d, A, d, g_A, d, lse, cu, bk, bv, B, H, d // H, window, window, float(p),
           seed_a, gQ, d, gKV, 2 * d, gKV[:, d:], 2 * d, delta, part)
        d_in_w, d_in_b = new(3 * d, d), new(3 * d)
        with _OnSide(dev, defer) as sd:
            side_ctxs.append(sd)
            sd.uses(gQ, gKV, q, x, d_in_w, d_in_b)
            _gemm(gQ, d, 0, q, d, 0, d_in_w, d, None, None, 0, d, d, M, 0, sp, d_in_b)
            _gemm(gKV, 2 * d, 0, x, d, 0, d_in_w[d:], d, None, None, 0, 2 * d, d, M, 0, sp, d_in_b[d:])
            if pad_keys:   # the pad keys' share: b_k has no gradient in total (it shifts every logit of a query alike), b_v gets theirs
                sd.uses(part)
                d_in_b[d:2 * d].zero_()
                d_in_b[2 * d:] += part.sum(0)
        g_q, g_kv = new(M, d), new(M, d)
        _gemm_group([(gQ, d, in_w, d, g_q, d, None, g_y, d, M, d, d, 0),
                     (gKV, 2 * d, in_w[d:], d, g_kv, d, None, None, 0, M, d, 2 * d, 0)], 1, 0)
        g_x, d_ln1w, d_ln1b = ln_bwd(g_q, x, ln1_w, mean1, rstd1, res=g_kv)     # g_x = LN1'(g_q) + gKV Wkv
        if side_ctxs:
            side_ctxs[-1].join_now()
        return (g_x, None, d_ln1w, d_ln1b, d_in_w, d_in_b, d_wo, d_bo, d_ln2w, d_ln2b, d_w1, d_b1, d_w2, d_b2, None)


_GRAD_OFFSETS: tp.Dict[tp.Tuple[int, int], tp.List[int]] = {}


class WeightPlanes:
    """The bf16 planes (exact three-way split, `rt_split_planes`) of ONE contiguous fp32 parameter range — every weight of a layer stack
    whose parameters live in a flat buffer (`lightning.FlatAdam`).  `refresh()` re-splits the range (one launch, a few microseconds:
    the range is a few MB) and is called at the start of every forward pass of the stack, so the planes can never be stale;
    `of(weight)` = the plane-0 pointer of a weight inside the range, or None."""

    def __init__(self, params: tp.Sequence[torch.Tensor]) -> None:
        ps = [p for p in params if p.is_cuda and p.dtype == torch.float32]
        self.ok = False
        if not ps:
            return
        base = ps[0].untyped_storage().data_ptr()
        if any(p.untyped_storage().data_ptr() != base or not p.is_contiguous() for p in ps):
            return      # parameters in separate allocations (no flat buffer): the stack runs without planes
        lo = min(p.data_ptr() for p in ps)
        hi = max(p.data_ptr() + p.numel() * 4 for p in ps)
        if lo % 16 != 0:
            return
        self.lo, self.n = lo, ((hi - lo) // 4 + 3) // 4 * 4
        end = base + ps[0].untyped_storage().nbytes()
        if lo + self.n * 4 > end:
            self.n = (end - lo) // 16 * 4
        self.stride = (self.n + 7) // 8 * 8
        self.planes = torch.empty((3 * self.stride,), dtype=torch.int16, device=ps[0].device)
        self.ok = True

    def refresh(self) -> None:
        if self.ok:
            _c("rt_split_planes", self.lo, self.n, self.planes, self.stride)

    def of(self, w: torch.Tensor) -> tp.Optional[int]:
        if not self.ok:
            return None
        off = w.data_ptr() - self.lo
        if off < 0 or off % 16 != 0 or off + w.numel() * 4 > self.n * 4:
            return None
        return self.planes.data_ptr() + off // 2


def weight_planes_enabled() -> bool:
    """Pre-split weight planes (K7w) are bf16x6 arithmetic: RT_GEMM_SPLIT=exact (every product on the f32-input MFMA) switches them off."""
    return os.environ.get("RT_WEIGHT_PLANES", "1") != "0" and os.environ.get("RT_GEMM_SPLIT", "") != "exact"


def _block_desc(rows: int, rows_real: int, cu: torch.Tensor, B: int, H: int, d: int, dff: int, window: int, pad_keys: bool, p: float,
                eps1: float, eps2: float, seeds: tp.Tuple[int, int, int, int, int], params: tp.Sequence[torch.Tensor],
                planes: tp.Optional[WeightPlanes] = None) -> "_lib.SasrecBlock":
    blk = _lib.SasrecBlock()
    if planes is not None and planes.ok and weight_planes_enabled():
        ptrs = [planes.of(params[i]) for i in (2, 4, 8, 10)]          # in_w, out_w, w1, w2
        if all(q is not None for q in ptrs):
            blk.in_wp, blk.out_wp, blk.w1_wp, blk.w2_wp = ptrs
            blk.wp_stride = planes.stride
    blk.rows, blk.rows_real, blk.B, blk.H, blk.d, blk.dff, blk.window, blk.pad_keys = rows, rows_real, B, H, d, dff, window, int(pad_keys)
    blk.p_drop, blk.eps1, blk.eps2 = float(p), float(eps1), float(eps2)
    blk.seed_attn, blk.seed_h, blk.sid_h, blk.seed_o, blk.sid_o = seeds
    blk.cu = cu.data_ptr()
    (blk.ln1_w, blk.ln1_b, blk.in_w, blk.in_b, blk.out_w, blk.out_b, blk.ln2_w, blk.ln2_b, blk.w1, blk.b1, blk.w2,
     blk.b2) = [t.data_ptr() for t in params]
    return blk


class _SASRecLayerPackedNative(torch.autograd.Function):
    """`_SASRecLayerPacked` issued by the native executor (`rt_sasrec_block_packed_fwd / _bwd`, csrc/rt_block.hip): ONE C call per
    direction instead of ~10 / ~18 ctypes launches with a torch.empty per buffer — same kernels, same order, same dropout streams
    (the node draws them from `RNG` exactly as the Python node does).  The activations live in one `saved` buffer, the data
    gradients in one scratch buffer, the twelve parameter gradients are views of one flat buffer."""

    @staticmethod
    def forward(ctx, x, cu, ln1_w, ln1_b, in_w, in_b, out_w, out_b, ln2_w, ln2_b, w1, b1, w2, b2, meta):
        import ctypes

        B, H, window, pad_keys, p, eps1, eps2, rows_real, planes = meta
        x = x.contiguous()
        M, d = x.shape
        dff = w1.shape[0]
        lib = _lib.load()
        seed_a, seed_h, seed_o = 0, (0, 0), (0, 0)
        if p > 0:       # the same draws, in the same order, as the Python node
            s0, sid = RNG.next()
            seed_a = (s0 + 0xD1B54A32D192ED03 * sid) & 0xFFFFFFFFFFFFFFFF
            seed_h = RNG.next()
            seed_o = RNG.next()
        params = (ln1_w, ln1_b, in_w, in_b, out_w, out_b, ln2_w, ln2_b, w1, b1, w2, b2)
        blk = _block_desc(M, rows_real, cu, B, H, d, dff, window, pad_keys, p, eps1, eps2, (seed_a, seed_h[0], seed_h[1], seed_o[0], seed_o[1]),
                          params, planes)
        saved = torch.empty((lib.rt_sasrec_block_saved_floats(M, d, dff, H, 1 if p > 0 else 0),), dtype=torch.float32, device=x.device)
        out = torch.empty((M, d), dtype=torch.float32, device=x.device)
        _c("rt_sasrec_block_packed_fwd", ctypes.addressof(blk), x, saved, out)
        ctx.save_for_backward(x, cu, saved, *params)
        ctx.blk_meta = (B, H, window, pad_keys, p, eps1, eps2, rows_real, (seed_a, seed_h[0], seed_h[1], seed_o[0], seed_o[1]), planes)
        return out

    @staticmethod
    def backward(ctx, g_out):
        import ctypes

        x, cu, saved, *params = ctx.saved_tensors
        B, H, window, pad_keys, p, eps1, eps2, rows_real, seeds, planes = ctx.blk_meta
        g_out = g_out.contiguous()
        M, d = g_out.shape
        dff = params[8].shape[0]
        dev = g_out.device
        lib = _lib.load()
        blk = _block_desc(M, rows_real, cu, B, H, d, dff, window, pad_keys, p, eps1, eps2, seeds, params, planes)
        key = (d, dff)
        offs = _GRAD_OFFSETS.get(key)
        if offs is None:
            arr = (ctypes.c_int64 * 13)()
            lib.rt_sasrec_block_grad_offsets(d, dff, arr)
            offs = _GRAD_OFFSETS[key] = list(arr)
        sp = _wgrad_splits(M)
        scratch_bytes = lib.rt_sasrec_block_bwd_scratch_bytes(M, B, d, dff, H, sp)
        scratch = torch.empty((scratch_bytes,), dtype=torch.uint8, device=dev)
        grads = torch.empty((offs[12],), dtype=torch.float32, device=dev)
        g_x = torch.empty((M, d), dtype=torch.float32, device=dev)
        use_side = _side_enabled() and _steals_grad(*params)
        _c("rt_sasrec_block_packed_bwd", ctypes.addressof(blk), x, saved, g_out, g_x, grads, scratch, scratch_bytes, sp, 1 if use_side else 0)
        if use_side:   # the side stream may read these until the join at the end of the backward pass (FlatAdam.step / callback)
            if not _NATIVE_KEEPALIVE:
                torch.autograd.Variable._execution_engine.queue_callback(join_side_streams)
            _NATIVE_KEEPALIVE.append((x, saved, g_out, scratch, grads))
        views = [grads[offs[i]:offs[i] + t.numel()].view_as(t) for i, t in enumerate(params)]
        return (g_x, None, *views, None)


class _PreLNLayerPackedNative(torch.autograd.Function):
    """One packed Pre-LN block issued by the native executor (`rt_preln_block_packed_fwd / _bwd`, csrc/rt_block.hip): ONE C call per
    direction instead of ~11 / ~20 launches from Python — same kernels, order and dropout streams as
    `nn.PreLNTransformerLayer.forward_packed` (the cross-check: tests/test_packed_bert_gpu.py)."""

    @staticmethod
    def _desc(M, cu, params, meta, seeds):
        B, H, window, causal, p, eps1, eps2, rows_real, planes = meta
        d, dff = params[2].shape[1], params[8].shape[0]
        blk = _lib.PreLNBlock()
        if planes is not None and planes.ok and weight_planes_enabled():
            ptrs = [planes.of(params[i]) for i in (2, 4, 8, 10)]
            if all(q is not None for q in ptrs):
                blk.in_wp, blk.out_wp, blk.w1_wp, blk.w2_wp = ptrs
                blk.wp_stride = planes.stride
        blk.rows, blk.rows_real, blk.B, blk.H, blk.d, blk.dff, blk.window, blk.causal = M, rows_real, B, H, d, dff, window, int(causal)
        blk.p_drop, blk.eps1, blk.eps2 = float(p), float(eps1), float(eps2)
        (blk.seed_attn, blk.seed1, blk.sid1, blk.seed_h, blk.sid_h, blk.seed2, blk.sid2, blk.seed3, blk.sid3) = seeds
        blk.cu = cu.data_ptr()
        (blk.ln1_w, blk.ln1_b, blk.in_w, blk.in_b, blk.out_w, blk.out_b, blk.ln2_w, blk.ln2_b, blk.w1, blk.b1, blk.w2,
         blk.b2) = [t.data_ptr() for t in params]
        return blk

    @staticmethod
    def forward(ctx, x, cu, ln1_w, ln1_b, in_w, in_b, out_w, out_b, ln2_w, ln2_b, w1, b1, w2, b2, meta):
        import ctypes

        p = meta[4]
        x = x.contiguous()
        M, d = x.shape
        dff = w1.shape[0]
        lib = _lib.load()
        seeds = (0,) * 9
        if p > 0:       # the draws of the Python block, in its order: attention, dropout after it, the feed-forward's, after it, dropout_3
            s0, sid = RNG.next()
            s1, sh, s2, s3 = RNG.next(), RNG.next(), RNG.next(), RNG.next()
            seeds = ((s0 + 0xD1B54A32D192ED03 * sid) & 0xFFFFFFFFFFFFFFFF, s1[0], s1[1], sh[0], sh[1], s2[0], s2[1], s3[0], s3[1])
        params = (ln1_w, ln1_b, in_w, in_b, out_w, out_b, ln2_w, ln2_b, w1, b1, w2, b2)
        blk = _PreLNLayerPackedNative._desc(M, cu, params, meta, seeds)
        saved = torch.empty((lib.rt_preln_block_saved_floats(M, d, dff, meta[1]),), dtype=torch.float32, device=x.device)
        out = torch.empty((M, d), dtype=torch.float32, device=x.device)
        _c("rt_preln_block_packed_fwd", ctypes.addressof(blk), x, saved, out)
        ctx.save_for_backward(x, cu, saved, *params)
        ctx.blk_meta = (meta, seeds)
        return out

    @staticmethod
    def backward(ctx, g_out):
        import ctypes

        x, cu, saved, *params = ctx.saved_tensors
        meta, seeds = ctx.blk_meta
        g_out = g_out.contiguous()
        M, d = g_out.shape
        dff = params[8].shape[0]
        dev = g_out.device
        lib = _lib.load()
        blk = _PreLNLayerPackedNative._desc(M, cu, params, meta, seeds)
        key = (d, dff)
        offs = _GRAD_OFFSETS.get(key)
        if offs is None:
            arr = (ctypes.c_int64 * 13)()
            lib.rt_sasrec_block_grad_offsets(d, dff, arr)
            offs = _GRAD_OFFSETS[key] = list(arr)
        sp = _wgrad_splits(M)
        scratch_bytes = lib.rt_preln_block_bwd_scratch_bytes(M, d, dff, meta[1], sp)
        scratch = torch.empty((scratch_bytes,), dtype=torch.uint8, device=dev)
        grads = torch.empty((offs[12],), dtype=torch.float32, device=dev)
        g_x = torch.empty((M, d), dtype=torch.float32, device=dev)
        use_side = _side_enabled() and _steals_grad(*params)
        _c("rt_preln_block_packed_bwd", ctypes.addressof(blk), x, saved, g_out, g_x, grads, scratch, scratch_bytes, sp, 1 if use_side else 0)
        if use_side:
            if not _NATIVE_KEEPALIVE:
                torch.autograd.Variable._execution_engine.queue_callback(join_side_streams)
            _NATIVE_KEEPALIVE.append((x, saved, g_out, scratch, grads))
        views = [grads[offs[i]:offs[i] + t.numel()].view_as(t) for i, t in enumerate(params)]
        return (g_x, None, *views, None)


def preln_layer_packed_train(x: torch.Tensor, cu: torch.Tensor, B: int, H: int, window: int, causal: bool, p: float,
                             ln1: tp.Tuple[torch.Tensor, torch.Tensor, float], in_proj: tp.Tuple[torch.Tensor, torch.Tensor],
                             out_proj: tp.Tuple[torch.Tensor, torch.Tensor], ln2: tp.Tuple[torch.Tensor, torch.Tensor, float],
                             ff1: tp.Tuple[torch.Tensor, torch.Tensor], ff2: tp.Tuple[torch.Tensor, torch.Tensor], rows_real: int,
                             planes: tp.Optional[WeightPlanes] = None) -> torch.Tensor:
    """One packed Pre-LN block with autograd through the native executor (biases everywhere, GELU feed-forward)."""
    return _PreLNLayerPackedNative.apply(_chk(x, "preln_layer_packed_train"), cu, ln1[0], ln1[1], in_proj[0], in_proj[1], out_proj[0],
                                         out_proj[1], ln2[0], ln2[1], ff1[0], ff1[1], ff2[0], ff2[1],
                                         (B, H, window, causal, p, ln1[2], ln2[2], int(rows_real), planes))


def native_block_enabled() -> bool:
    return os.environ.get("RT_NATIVE_BLOCK", "1") != "0"


def sasrec_layer_packed_train(x: torch.Tensor, cu: torch.Tensor, B: int, H: int, window: int, pad_keys: bool, p: float,
                              ln1: tp.Tuple[torch.Tensor, torch.Tensor, float], in_proj: tp.Tuple[torch.Tensor, torch.Tensor],
                              out_proj: tp.Tuple[torch.Tensor, torch.Tensor], ln2: tp.Tuple[torch.Tensor, torch.Tensor, float],
                              ff1: tp.Tuple[torch.Tensor, torch.Tensor], ff2: tp.Tuple[torch.Tensor, torch.Tensor],
                              rows_real: tp.Optional[int] = None, planes: tp.Optional[WeightPlanes] = None) -> torch.Tensor:
    """One packed SASRec block with autograd.  rows_real (the number of rows that belong to sessions; host-side knowledge of the
    caller) selects the native executor; without it the Python node issues the same kernels one by one."""
    if native_block_enabled() and rows_real is not None and x.shape[0] % 128 == 0:
        return _SASRecLayerPackedNative.apply(_chk(x, "sasrec_layer_packed_train"), cu, ln1[0], ln1[1], in_proj[0], in_proj[1], out_proj[0],
                                              out_proj[1], ln2[0], ln2[1], ff1[0], ff1[1], ff2[0], ff2[1],
                                              (B, H, window, pad_keys, p, ln1[2], ln2[2], int(rows_real), planes))
    return _SASRecLayerPacked.apply(_chk(x, "sasrec_layer_packed_train"), cu, ln1[0], ln1[1], in_proj[0], in_proj[1], out_proj[0],
                                    out_proj[1], ln2[0], ln2[1], ff1[0], ff1[1], ff2[0], ff2[1],
                                    (B, H, window, pad_keys, p, ln1[2], ln2[2]))


def sasrec_layer_last(x: torch.Tensor, ids: torch.Tensor, B: int, L: int, H: int, causal: bool, keypad: bool,
                      ln1: tp.Tuple[torch.Tensor, torch.Tensor, float], in_proj: tp.Tuple[torch.Tensor, torch.Tensor],
                      out_proj: tp.Tuple[torch.Tensor, torch.Tensor], ln2: tp.Tuple[torch.Tensor, torch.Tensor, float],
                      ff1: tp.Tuple[torch.Tensor, torch.Tensor], ff2: tp.Tuple[torch.Tensor, torch.Tensor]) -> torch.Tensor:
    """Inference only: the SASRec block's output at the LAST position of every session, [B, d].

    recommend() keeps `session_embs[:, -1, :]` (lightning.py:393-397); in the final block only the key / value projection
    needs every position — the query projection, the attention (`rt_mha_last_fwd`), out_proj and the feed-forward run on
    one row per session.  Row for row the same arithmetic as `sasrec_layer` (the GEMM's k order does not depend on M)."""
    x = _chk(x, "sasrec_layer_last").contiguous()
    M, d = x.shape
    dev = x.device
    new = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)  # noqa: E731
    ids_flat = ids.reshape(-1)
    in_w, in_b = in_proj
    x0 = new(M, d)
    _c("rt_mul_mask", x, None, ids_flat, d, x.numel(), x0)
    KV = new(M, 2 * d)
    _gemm(x0, d, 1, in_w[d:], d, 1, KV, 2 * d, in_b[d:], None, 0, M, 2 * d, d)
    x0_last = x0.view(B, L, d)[:, L - 1, :].contiguous()
    q, mean, rstd = new(B, d), new(B), new(B)
    _c("rt_layernorm_fwd", x0_last, ln1[0], ln1[1], float(ln1[2]), B, d, q, mean, rstd)
    Q = new(B, d)
    _gemm(q, d, 1, in_w, d, 1, Q, d, in_b, None, 0, B, d, d)
    A = new(B, d)
    _c("rt_mha_last_fwd", Q, d, KV, 2 * d, KV[:, d:], 2 * d, ids_flat, B, H, L, d // H, int(causal), int(keypad), A, d)
    y = new(B, d)
    _gemm(A, d, 1, out_proj[0], d, 1, y, d, out_proj[1], q, d, B, d, d)
    f = new(B, d)
    _c("rt_layernorm_fwd", y, ln2[0], ln2[1], float(ln2[2]), B, d, f, mean, rstd)
    dff = ff1[0].shape[0]
    h = new(B, dff)
    _gemm(f, d, 1, ff1[0], d, 1, h, dff, ff1[1], None, 0, B, dff, d, 1)
    out = new(B, d)
    _gemm(h, dff, 1, ff2[0], dff, 1, out, d, ff2[1], f, d, B, d, dff)
    return out


class _MHAVarlen(torch.autograd.Function):
    """Causal softmax attention over PACKED sessions with the window's pad keys as one virtual key (`rt_mha_varlen_train_fwd` /
    `rt_mha_varlen_bwd`).  q [Np, d]; kv [Np, 2d] (k | v column blocks); cu [B+1]; bk / bv [d] or None (pad keys masked).
    Gradients: dq, dkv, and for bk / bv the pad keys' share (the real rows' share arrives through the projection's bias
    gradient; for bk the two cancel exactly)."""

    @staticmethod
    def forward(ctx, q, kv, bk, bv, cu, B, H, window, p):
        Np, d = q.shape
        hd = d // H
        o = torch.zeros((Np, d), dtype=torch.float32, device=q.device)           # rows behind the last session stay zero
        lse = torch.zeros((Np, H), dtype=torch.float32, device=q.device)
        seed = 0
        if p > 0:
            s0, sid = RNG.next()
            seed = (s0 + 0xD1B54A32D192ED03 * sid) & 0xFFFFFFFFFFFFFFFF
        _c("rt_mha_varlen_train_fwd", q, d, kv, 2 * d, kv[:, d:], 2 * d, cu, bk, bv, B, H, hd, window, window, float(p), seed, o, d, lse)
        ctx.save_for_backward(q, kv, bk, bv, cu, o, lse)
        ctx.meta = (B, H, window, p, seed)
        return o

    @staticmethod
    def backward(ctx, do):
        q, kv, bk, bv, cu, o, lse = ctx.saved_tensors
        B, H, window, p, seed = ctx.meta
        Np, d = q.shape
        hd = d // H
        do = do.contiguous()
        dq, dkv = torch.zeros_like(q), torch.zeros_like(kv)
        delta = torch.empty((Np, H), dtype=torch.float32, device=q.device)
        part = torch.empty((B, d), dtype=torch.float32, device=q.device) if bv is not None else None
        _c("rt_mha_varlen_bwd", q, d, kv, 2 * d, kv[:, d:], 2 * d, o, d, do, d, lse, cu, bk, bv, B, H, hd, window, window, float(p), seed,
           dq, d, dkv, 2 * d, dkv[:, d:], 2 * d, delta, part)
        # the pad keys' share of d_bk: the key bias shifts every logit of a query alike, so the gradient summed over ALL keys of
        # the window is zero — the pads carry exactly minus what the real key rows carry
        dbk = -dkv[:, :d].sum(0) if bk is not None else None
        dbv = part.sum(0) if part is not None else None
        return dq, dkv, dbk, dbv, None, None, None, None, None


def mha_varlen(q: torch.Tensor, kv: torch.Tensor, bk: tp.Optional[torch.Tensor], bv: tp.Optional[torch.Tensor], cu: torch.Tensor, B: int,
               H: int, window: int, p: float) -> torch.Tensor:
    return _MHAVarlen.apply(_chk(q, "mha_varlen").contiguous(), _chk(kv, "mha_varlen").contiguous(), bk, bv, cu, B, H, window, p)


class _MHAVarlenQKV(torch.autograd.Function):
    """Softmax attention over PACKED sessions from one packed projection qkv [Np, 3d] (q | k | v column blocks, the layout
    `nn.MultiheadAttention`'s in_proj produces), key-padding-masked windows (no pad keys).  causal: `rt_mha_varlen_train_fwd` /
    `_bwd`; bidirectional (BERT4Rec, bert4rec.py:200): `rt_mha_varlen_bidir_fwd` / `_bwd`.  cu [B+1] (the unused tail of the row block
    may ride along as one more session, see `TransformerTorchBackbone.encode_packed_train`)."""

    @staticmethod
    def forward(ctx, qkv, cu, B, H, window, causal, p, covers_all_rows, n_prefixed=None):
        Np, d3 = qkv.shape
        d = d3 // 3
        hd = d // H
        alloc = torch.empty if covers_all_rows else torch.zeros                 # rows behind the last session stay zero
        o = alloc((Np, d), dtype=torch.float32, device=qkv.device)
        lse = alloc((Np, H), dtype=torch.float32, device=qkv.device)
        seed = 0
        if p > 0:
            s0, sid = RNG.next()
            seed = (s0 + 0xD1B54A32D192ED03 * sid) & 0xFFFFFFFFFFFFFFFF
        q, k, v = qkv, qkv[:, d:], qkv[:, 2 * d:]
        if n_prefixed is not None:       # sessions 0 .. n_prefixed - 1 behind the shared pad prefix (session n_prefixed): causal only
            _c("rt_mha_varlen_prefix_fwd", q, d3, k, d3, v, d3, cu, B, int(n_prefixed), H, hd, window, window, float(p), seed, o, d, lse)
        elif causal:
            _c("rt_mha_varlen_train_fwd", q, d3, k, d3, v, d3, cu, None, None, B, H, hd, window, window, float(p), seed, o, d, lse)
        else:
            _c("rt_mha_varlen_bidir_fwd", q, d3, k, d3, v, d3, cu, B, H, hd, window, float(p), seed, o, d, lse)
        ctx.save_for_backward(qkv, cu, o, lse)
        ctx.meta = (B, H, window, causal, p, seed, covers_all_rows, n_prefixed)
        return o

    @staticmethod
    def backward(ctx, do):
        qkv, cu, o, lse = ctx.saved_tensors
        B, H, window, causal, p, seed, covers_all_rows, n_prefixed = ctx.meta
        Np, d3 = qkv.shape
        d = d3 // 3
        hd = d // H
        do = do.contiguous()
        dqkv = (torch.empty_like if covers_all_rows else torch.zeros_like)(qkv)
        delta = torch.empty((Np, H), dtype=torch.float32, device=qkv.device)
        q, k, v = qkv, qkv[:, d:], qkv[:, 2 * d:]
        if n_prefixed is not None:
            ws_bytes = _lib.load().rt_mha_varlen_prefix_bwd_workspace_bytes(window, H, hd)
            ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=qkv.device)
            _c("rt_mha_varlen_prefix_bwd", q, d3, k, d3, v, d3, o, d, do, d, lse, cu, B, int(n_prefixed), H, hd, window, window, float(p), seed,
               dqkv, d3, dqkv[:, d:], d3, dqkv[:, 2 * d:], d3, delta, ws, ws_bytes)
        elif causal:
            _c("rt_mha_varlen_bwd", q, d3, k, d3, v, d3, o, d, do, d, lse, cu, None, None, B, H, hd, window, window, float(p), seed,
               dqkv, d3, dqkv[:, d:], d3, dqkv[:, 2 * d:], d3, delta, None)
        else:
            _c("rt_mha_varlen_bidir_bwd", q, d3, k, d3, v, d3, o, d, do, d, lse, cu, B, H, hd, window, float(p), seed,
               dqkv, d3, dqkv[:, d:], d3, dqkv[:, 2 * d:], d3, delta)
        return dqkv, None, None, None, None, None, None, None, None


def mha_varlen_qkv(qkv: torch.Tensor, cu: torch.Tensor, B: int, H: int, window: int, causal: bool, p: float,
                   covers_all_rows: bool = False, n_prefixed: tp.Optional[int] = None) -> torch.Tensor:
    """n_prefixed: session number `n_prefixed` of cu is the SHARED PAD PREFIX (the window's `window` positions as pad rows, carried once
    per batch) and sessions 0 .. n_prefixed - 1 see its first window - n_b rows as keys in front of their own — a causal stack that
    neither masks pad keys nor re-zeroes pad rows (LiGR: the reference's default eSASRec) on packed rows (`rt_mha_varlen_prefix_*`)."""
    return _MHAVarlenQKV.apply(_chk(qkv, "mha_varlen_qkv").contiguous(), cu, B, H, window, bool(causal), p, bool(covers_all_rows), n_prefixed)


def mha_varlen_qkv_infer(qkv: torch.Tensor, cu: torch.Tensor, B: int, H: int, window: int, causal: bool) -> torch.Tensor:
    """Inference twin of `mha_varlen_qkv` (no lse, no dropout)."""
    Np, d3 = qkv.shape
    d = d3 // 3
    o = torch.zeros((Np, d), dtype=torch.float32, device=qkv.device)
    if causal:
        _c("rt_mha_varlen_fwd", qkv, d3, qkv[:, d:], d3, qkv[:, 2 * d:], d3, cu, None, None, B, H, d // H, window, window, o, d)
    else:
        _c("rt_mha_varlen_bidir_fwd", qkv, d3, qkv[:, d:], d3, qkv[:, 2 * d:], d3, cu, B, H, d // H, window, 0.0, 0, o, d, None)
    return o


def mha_bidir_supported(n_heads: int, d: int, window: int) -> bool:
    """`rt_mha_varlen_bidir_*` (the streamed bf16-plane kernels, K4v3): head size 32 / 64 / 128, any window (the partner rows stream
    through 48 KB of LDS; the whole-session-image kernels of round 3 stopped where two images filled the 160 KB)."""
    return d % n_heads == 0 and d // n_heads in (32, 64, 128)


def mha_varlen_supported(n_heads: int, d: int, window: int) -> bool:
    """`rt_mha_varlen_fwd / _train_fwd / _bwd`: head size 32 / 64 / 128, any window (K4v3)."""
    return d % n_heads == 0 and d // n_heads in (32, 64, 128)


def sasrec_layer_packed(x: torch.Tensor, cu: torch.Tensor, B: int, H: int, window: int, pad_keys: bool, last_rows: tp.Optional[torch.Tensor],
                        ln1: tp.Tuple[torch.Tensor, torch.Tensor, float], in_proj: tp.Tuple[torch.Tensor, torch.Tensor],
                        out_proj: tp.Tuple[torch.Tensor, torch.Tensor], ln2: tp.Tuple[torch.Tensor, torch.Tensor, float],
                        ff1: tp.Tuple[torch.Tensor, torch.Tensor], ff2: tp.Tuple[torch.Tensor, torch.Tensor],
                        rows_real: tp.Optional[int] = None, planes: tp.Optional[WeightPlanes] = None,
                        kv_in: tp.Optional[torch.Tensor] = None, q_in: tp.Optional[torch.Tensor] = None,
                        Q_in: tp.Optional[torch.Tensor] = None) -> torch.Tensor:
    """Inference only: one causal SASRec block (sasrec.py:186-231) over PACKED sessions — `x` [Np, d] holds the real positions
    only (session b = rows cu[b] .. cu[b+1]-1, oldest first; Np = the row count rounded up to the 128-row GEMM tile, tail rows
    arbitrary).  The pad keys the reference's left-padded window shows to every query enter as one virtual key per query
    (`rt_mha_varlen_fwd`; `pad_keys` = the block runs without key-padding masks).  last_rows None: the block's output for
    every row, [Np, d]; last_rows [B] (= cu[1:] - 1): only the last position of every session, [B, d] — no product over all rows is
    left then (`mha_varlen_last_x`).  kv_in [Np, 2d] (with last_rows None): the block's keys | values handed in (the first block of
    recommend(): a gather from two projected tables, `nn.TransformerTorchBackbone.encode_last_packed`); q_in / Q_in [Np, d] (with kv_in):
    LN1(x) and the projected queries handed in as well (`rt_embed_block1_fwd`) — `x` is not read then."""
    x = _chk(x, "sasrec_layer_packed").contiguous()
    Np, d = x.shape
    dev = x.device
    new = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)  # noqa: E731
    in_w, in_b = in_proj
    hd = d // H
    if native_block_enabled() and rows_real is not None and Np % 128 == 0 and ff1[1] is not None and ff2[1] is not None:
        import ctypes   # the same launch sequence as below, issued by the native executor (csrc/rt_block.hip): ONE call per block

        lib = _lib.load()
        dff = ff1[0].shape[0]
        blk = _block_desc(Np, int(rows_real), cu, B, H, d, dff, window, pad_keys, 0.0, ln1[2], ln2[2], (0, 0, 0, 0, 0),
                          (ln1[0], ln1[1], in_w, in_b, out_proj[0], out_proj[1], ln2[0], ln2[1], ff1[0], ff1[1], ff2[0], ff2[1]), planes)
        last = last_rows is not None
        scratch = new(lib.rt_sasrec_block_infer_scratch_floats(Np, B, d, dff, 1 if last else 0))
        out = new(B if last else Np, d)
        pre = not last and kv_in is not None and q_in is not None and Q_in is not None
        _c("rt_sasrec_block_packed_infer", ctypes.addressof(blk), x, q_in if pre else None, Q_in if pre else None,
           kv_in if not last else None, last_rows, scratch, out)
        return out
    bk = in_b[d:2 * d] if pad_keys else None
    bv = in_b[2 * d:] if pad_keys else None
    mean, rstd = new(Np), new(Np)
    if last_rows is None:
        if kv_in is not None and q_in is not None and Q_in is not None:
            q, Q, KV = q_in, Q_in, kv_in
        elif kv_in is not None:
            q = new(Np, d)
            _c("rt_layernorm_fwd", x, ln1[0], ln1[1], float(ln1[2]), Np, d, q, mean, rstd)
            Q, KV = new(Np, d), kv_in
            _gemm(q, d, 1, in_w, d, 1, Q, d, in_b, None, 0, Np, d, d)
        else:
            q = new(Np, d)
            _c("rt_layernorm_fwd", x, ln1[0], ln1[1], float(ln1[2]), Np, d, q, mean, rstd)
            Q, KV = new(Np, d), new(Np, 2 * d)
            _gemm_group([(q, d, in_w, d, Q, d, in_b, None, 0, Np, d, d, 0),
                         (x, d, in_w[d:], d, KV, 2 * d, in_b[d:], None, 0, Np, 2 * d, d, 0)], 1, 1)
        A = torch.zeros((Np, d), dtype=torch.float32, device=dev)          # tail rows are not written by the kernel
        _c("rt_mha_varlen_fwd", Q, d, KV, 2 * d, KV[:, d:], 2 * d, cu, bk, bv, B, H, hd, window, window, A, d)
        rows = Np
    else:
        x_last = x.index_select(0, last_rows)
        rows = B
        q = new(B, d)
        _c("rt_layernorm_fwd", x_last, ln1[0], ln1[1], float(ln1[2]), B, d, q, mean, rstd)
        Q = new(B, d)
        _gemm(q, d, 1, in_w, d, 1, Q, d, in_b, None, 0, B, d, d)
        if mha_varlen_last_x_supported(d, H):      # no key / value rows: one pass over the block input (as the native executor)
            A = mha_varlen_last_x(Q, x, in_w, in_b, cu, B, H, window, pad_keys)
        else:
            KV = new(Np, 2 * d)
            _gemm(x, d, 1, in_w[d:], d, 1, KV, 2 * d, in_b[d:], None, 0, Np, 2 * d, d)
            A = new(B, d)
            _c("rt_mha_varlen_last_fwd", Q, d, KV, 2 * d, KV[:, d:], 2 * d, cu, bk, bv, B, H, hd, window, window, A, d)
    y = new(rows, d)
    _gemm(A, d, 1, out_proj[0], d, 1, y, d, out_proj[1], q, d, rows, d, d)
    f = new(rows, d)
    _c("rt_layernorm_fwd", y, ln2[0], ln2[1], float(ln2[2]), rows, d, f, mean, rstd)
    dff = ff1[0].shape[0]
    h = new(rows, dff)
    _gemm(f, d, 1, ff1[0], d, 1, h, dff, ff1[1], None, 0, rows, dff, d, 1)
    out = new(rows, d)
    _gemm(h, dff, 1, ff2[0], dff, 1, out, d, ff2[1], f, d, rows, d, dff)
    return out


def mha_varlen_last_x_supported(d: int, H: int) -> bool:
    """Does `mha_varlen_last_x` serve this width (`rt_mha_varlen_last_x_fwd`: d in {64, 128, 256, 512}, whole 8-column head groups)?"""
    return d in (64, 128, 256, 512) and H > 0 and d % H == 0 and (d // H) % 8 == 0 and H <= d // 32


def mha_varlen_last_x(q_last: torch.Tensor, x_rows: torch.Tensor, in_w: torch.Tensor, in_b: torch.Tensor, cu: torch.Tensor, B: int, H: int,
                      window: int, pad_keys: bool, prefix_row: int = -1) -> torch.Tensor:
    """Inference: the attention output [B, d] of the LAST query of every packed session WITHOUT projecting keys / values for the rows
    (`rt_mha_varlen_last_x_fwd`, include/rectools_hip.h): q_last [B, d] = the projected queries, x_rows [Np, d] = what the key / value
    projection would read (the block input, or its LayerNorm), in_w / in_b = the packed in_proj parameters [3d, d] / [3d].
    q.(W_k x_j + b_k) = (W_k^T q).x_j + const and sum_j p_j (W_v x_j + b_v) = W_v (sum_j p_j x_j) + b_v: two [B, .] products over
    head-expanded weights around one pass over x_rows.  prefix_row >= 0: the shared pad prefix starts at that row of x_rows
    (`LiGRLayers.packed_mode() == "prefix"`)."""
    d = int(x_rows.shape[1])
    dev = x_rows.device
    new = lambda *sh: torch.empty(sh, dtype=torch.float32, device=dev)  # noqa: E731
    Ek, Ev = new(d, H * d), new(d, H * d)
    _c("rt_mha_last_x_expand", in_w[d:2 * d], d, H, Ek)
    _c("rt_mha_last_x_expand", in_w[2 * d:], d, H, Ev)
    qk, xbar, A = new(B, H * d), new(B, H * d), new(B, d)
    _gemm(q_last, int(q_last.stride(0)), 1, Ek, H * d, 0, qk, H * d, None, None, 0, B, H * d, d)
    _c("rt_mha_varlen_last_x_fwd", qk, x_rows, int(x_rows.stride(0)), cu, B, H, d, window, window, 1 if pad_keys else 0, int(prefix_row), xbar)
    _gemm(xbar, H * d, 1, Ev, H * d, 1, A, d, in_b[2 * d:], None, 0, B, d, H * d)
    return A


HSTU_BUCKETS = 147      # csrc NBUCK: every bucket an int64 difference can reach (ln(2^63) / 0.301 = 145.08 -> buckets 0 .. 145, and one spare)


def hstu_time_thresholds(num_buckets: int = 128) -> torch.Tensor:
    """[148] int64: thr[b] (b < 147) = smallest |dt| >= 0 whose UNCLAMPED reference bucket trunc(log(max(1,|dt|)) / 0.301) is >= b, and behind
    them the number of entries of the model's `time_weights` the kernels may index (num_buckets + 1, at most 146).  The kernels find the unclamped bucket and read
    the weight of min(bucket, num_buckets): the reference's clamp(bucket, 0, num_buckets) (hstu.py:84-86) for ANY `num_buckets`.

    Computed with the reference's own float32 torch ops so that bucket edges are bit-identical.
    """
    top = HSTU_BUCKETS - 1
    i64max = torch.iinfo(torch.int64).max

    def bucket(x: int) -> int:
        return int(torch.clamp((torch.log(torch.abs(torch.tensor([x])).clamp(min=1)) / 0.301).long(), 0, top)[0])

    reach = bucket(i64max)       # 145: the last bucket any difference falls in
    thr = torch.zeros(HSTU_BUCKETS + 1, dtype=torch.int64)
    for b in range(1, top + 1):
        if b > reach:
            thr[b] = i64max
            continue
        guess = math.exp(0.301 * b)
        lo, hi = max(1, int(guess * 0.5)), min(int(guess * 2.0) + 2, i64max)
        while lo < hi:  # smallest x in [lo, hi] with bucket(x) >= b (bucket is monotone in x)
            mid = (lo + hi) // 2
            if bucket(mid) >= b:
                hi = mid
            else:
                lo = mid + 1
        thr[b] = lo
    thr[HSTU_BUCKETS] = min(int(num_buckets), reach) + 1      # (the spare bucket 146 — |dt| = 2^63 - 1 itself — reads the weight of 145)
    return thr


class _HstuAttn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, time_w, pos_w, ids, ts, thr, B, H, L):
        d = q.shape[1]
        hd = d // H
        o = torch.empty((B * L, d), dtype=torch.float32, device=q.device)
        _c("rt_hstu_attn_fwd", q, q.stride(0), k, k.stride(0), v, v.stride(0), ids, ts if time_w is not None else None,
           time_w, thr if time_w is not None else None, pos_w, B, H, L, hd, o, d)
        ctx.save_for_backward(q, k, v, time_w, pos_w, ids, ts, thr)
        ctx.meta = (B, H, L, hd)
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, time_w, pos_w, ids, ts, thr = ctx.saved_tensors
        B, H, L, hd = ctx.meta
        d = H * hd
        do = do.contiguous()
        dqkv = torch.empty((3, B * L, d), dtype=torch.float32, device=do.device)
        dtw = None if time_w is None else torch.zeros_like(time_w)
        dpw = None if pos_w is None else torch.zeros_like(pos_w)
        _c("rt_hstu_attn_bwd", q, q.stride(0), k, k.stride(0), v, v.stride(0), do, d, ids,
           ts if time_w is not None else None, time_w, thr if time_w is not None else None, pos_w, B, H, L, hd,
           dqkv[0], d, dqkv[1], d, dqkv[2], d, dtw, dpw)
        return dqkv[0], dqkv[1], dqkv[2], dtw, dpw, None, None, None, None, None, None


class _HstuAttnVarlen(torch.autograd.Function):
    """HSTU attention over PACKED sessions (`rt_hstu_attn_varlen_fwd` / `_bwd`): q / k / v [Np, d] (column blocks of one projection are
    fine: row strides are passed on), cu [B+1], ts [cu[B] + B] (n + 1 timestamps per session) or None.  Rows behind cu[B] stay zero."""

    @staticmethod
    def forward(ctx, q, k, v, time_w, pos_w, cu, ts, thr, B, H, window):
        Np, d = q.shape
        hd = d // H
        o = torch.zeros((Np, d), dtype=torch.float32, device=q.device)
        _c("rt_hstu_attn_varlen_fwd", q, q.stride(0), k, k.stride(0), v, v.stride(0), cu, ts if time_w is not None else None,
           time_w, thr if time_w is not None else None, pos_w, B, H, window, hd, o, d)
        ctx.save_for_backward(q, k, v, time_w, pos_w, cu, ts, thr)
        ctx.meta = (B, H, window, hd)
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, time_w, pos_w, cu, ts, thr = ctx.saved_tensors
        B, H, window, hd = ctx.meta
        Np, d = q.shape
        do = do.contiguous()
        dqkv = torch.zeros((3, Np, d), dtype=torch.float32, device=do.device)
        dtw = None if time_w is None else torch.zeros_like(time_w)
        dpw = None if pos_w is None else torch.zeros_like(pos_w)
        _c("rt_hstu_attn_varlen_bwd", q, q.stride(0), k, k.stride(0), v, v.stride(0), do, d, cu,
           ts if time_w is not None else None, time_w, thr if time_w is not None else None, pos_w, B, H, window, hd,
           dqkv[0], d, dqkv[1], d, dqkv[2], d, dtw, dpw)
        return dqkv[0], dqkv[1], dqkv[2], dtw, dpw, None, None, None, None, None, None


def hstu_attn_varlen(q, k, v, time_w, pos_w, cu, ts, thr, B: int, H: int, window: int) -> torch.Tensor:
    return _HstuAttnVarlen.apply(q, k, v, time_w, pos_w, cu, ts, thr, B, H, window)


def hstu_varlen_supported(n_heads: int, hd: int, window: int) -> bool:
    """The ring kernels behind `rt_hstu_attn_varlen_*`: head size 32 / 64; their LDS (4-stage ring + flags + bias tables) fits any window
    the position table allows here."""
    return hd in (32, 64) and window <= 2048


def collate_packed_ts(offsets: torch.Tensor, unix_ts: torch.Tensor, idx: torch.Tensor, cu: torch.Tensor, n_rows: int,
                      ctx: tp.Optional[torch.Tensor] = None) -> torch.Tensor:
    """`rt_collate_packed_ts`: the n + 1 timestamps of every session of a packed batch, [n_rows + B] (session b at cu[b] + b); training:
    the session's last n + 1, recommend (`ctx` [B]): its last n items' and the request's."""
    B = int(idx.numel())
    out = torch.empty((n_rows + B,), dtype=torch.int64, device=offsets.device)
    _c("rt_collate_packed_ts", offsets, unix_ts, idx, cu, ctx, B, n_rows + B, out)
    return out


def hstu_attn(q, k, v, time_w, pos_w, ids, ts, thr, B: int, H: int, L: int) -> torch.Tensor:
    return _HstuAttn.apply(q, k, v, time_w, pos_w, ids.reshape(-1), ts, thr, B, H, L)


class _STULayer(torch.autograd.Function):
    """One STU block (hstu.py:225-295) as a single autograd node.

        x0 = x * m;  n = LN_in(x0) * m;  [u v q k] = silu(n P);  a = hstu_attn(q, k, v, rab);  a' = dropout(a)
        o = dropout(u * LN_attn(a') * m);  out = o Wo^T + bo + x0            (m = the row mask `ids != 0`)

    Same kernels as the modular path (`nn.STULayer.forward_modular`); what the node adds is the backward data flow: the
    gradients of u, v, q, k are WRITTEN into the column slices of one [M, 4 H hd] buffer (autograd's SliceBackward built four
    zero-filled buffers of that size, copied a slice into each and added them: 0.95 + 0.34 + 0.28 ms of at:: kernels per C4
    step), u is read in place through a row stride, and the mask / skip-connection passes around LN_in ride inside its
    backward kernel (`rt_layernorm_bwd_fused`).

    Packed sessions (`meta` carries cu / rows_real, ids is None; `stu_layer_packed`): no pad rows, so m = 1 everywhere — the mask
    passes drop out, the attention is `rt_hstu_attn_varlen_*`, and the rows behind the last session are kept at zero where a weight
    gradient sums over all rows.
    """

    @staticmethod
    def forward(ctx, x, ids, ts, thr, ln1_w, ln1_b, uvqk_p, tw, pw, ln2_w, ln2_b, out_w, out_b, meta):
        B, L, H, hd, p_attn, p_mlp, eps1, eps2 = meta[:8]
        cu, rows_real = (meta[8], meta[9]) if len(meta) > 8 else (None, None)
        x = x.contiguous()
        M, d = x.shape
        hh = H * hd
        dev = x.device
        new = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)  # noqa: E731
        if cu is None:
            x0 = new(M, d)
            _c("rt_mul_mask", x, None, ids, d, x.numel(), x0)
        else:
            x0 = x
        n1, mean1, rstd1 = new(M, d), new(M), new(M)
        _c("rt_layernorm_fwd", x0, ln1_w, ln1_b, float(eps1), M, d, n1, mean1, rstd1)
        if cu is None:
            _c("rt_mul_mask", n1, None, ids, d, n1.numel(), n1)          # in place: n1 = LN(x0) * m
        z = new(M, 4 * hh)
        wp_p, wp_o = _planes_of(uvqk_p), _planes_of(out_w)
        _gemm_w(n1, d, uvqk_p, 4 * hh, 0, wp_p, z, 4 * hh, None, None, 0, M, 4 * hh, d)
        uvqk = new(M, 4 * hh)
        _c("rt_act_dropout_fwd", z, ACT_SILU, 0.0, 0, 0, z.numel(), None, uvqk)
        attn = new(M, hh)
        has_t = tw is not None
        if cu is None:
            _c("rt_hstu_attn_fwd", uvqk[:, 2 * hh:], 4 * hh, uvqk[:, 3 * hh:], 4 * hh, uvqk[:, hh:], 4 * hh, ids, ts if has_t else None,
               tw, thr if has_t else None, pw, B, H, L, hd, attn, hh)
        else:
            if rows_real < M:
                attn[rows_real:].zero_()                      # rows behind the last session: finite zeros
            _c("rt_hstu_attn_varlen_fwd", uvqk[:, 2 * hh:], 4 * hh, uvqk[:, 3 * hh:], 4 * hh, uvqk[:, hh:], 4 * hh, cu, ts if has_t else None,
               tw, thr if has_t else None, pw, B, H, L, hd, attn, hh)
        seed_a = seed_m = (0, 0)
        attn_d = attn
        if p_attn > 0:
            seed_a = RNG.next()
            attn_d = new(M, hh)
            _c("rt_act_dropout_fwd", attn, ACT_NONE, float(p_attn), seed_a[0], seed_a[1], attn.numel(), None, attn_d)
        la, mean2, rstd2 = new(M, hh), new(M), new(M)
        _c("rt_layernorm_fwd", attn_d, ln2_w, ln2_b, float(eps2), M, hh, la, mean2, rstd2)
        o_in = new(M, hh)
        _c("rt_mul_mask_ld", uvqk, 4 * hh, la, hh, ids, M, hh, o_in, hh)      # u * LN(a') * m, u read in place
        o_d = o_in
        if p_mlp > 0:
            seed_m = RNG.next()
            o_d = new(M, hh)
            _c("rt_act_dropout_fwd", o_in, ACT_NONE, float(p_mlp), seed_m[0], seed_m[1], o_in.numel(), None, o_d)
        out = new(M, d)
        _gemm_w(o_d, hh, out_w, hh, 1, wp_o, out, d, out_b, x0, d, M, d, hh)
        ctx.wp = (wp_p, wp_o)
        ctx.save_for_backward(ids, ts, thr, x0, n1, z, uvqk, attn_d, la, o_d, mean1, rstd1, mean2, rstd2, ln1_w, uvqk_p, tw, pw,
                              ln2_w, out_w, cu)
        ctx.meta = (B, L, H, hd, p_attn, p_mlp, seed_a, seed_m, rows_real)
        return out

    @staticmethod
    def backward(ctx, g_out):
        (ids, ts, thr, x0, n1, z, uvqk, attn_d, la, o_d, mean1, rstd1, mean2, rstd2, ln1_w, uvqk_p, tw, pw, ln2_w,
         out_w, cu) = ctx.saved_tensors
        B, L, H, hd, p_attn, p_mlp, seed_a, seed_m, rows_real = ctx.meta
        g_out = g_out.contiguous()
        M, d = g_out.shape
        hh = H * hd
        dev = g_out.device
        new = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)  # noqa: E731
        sp = _wgrad_splits(M)
        defer = _steals_grad(ln1_w, uvqk_p, ln2_w, out_w)
        lib = _lib.load()
        side = []
        # ---- out = o_d Wo^T + bo + x0
        d_wo, d_bo = new(d, hh), new(d)
        with _OnSide(dev, defer) as sd:
            side.append(sd)
            sd.uses(g_out, o_d, d_wo, d_bo)
            _gemm(g_out, d, 0, o_d, hh, 0, d_wo, hh, None, None, 0, d, hh, M, 0, sp, d_bo)
        g_od = new(M, hh)
        _gemm_w(g_out, d, out_w, hh, 0, ctx.wp[1], g_od, hh, None, None, 0, M, hh, d)
        if p_mlp > 0:
            g_oin = new(M, hh)
            _c("rt_act_dropout_bwd", g_od, g_od, ACT_NONE, float(p_mlp), seed_m[0], seed_m[1], g_od.numel(), g_oin)
        else:
            g_oin = g_od
        # ---- o_in = u * la * m : the u gradient goes straight into its column slice of the packed gradient
        g_uvqk = new(M, 4 * hh)
        _c("rt_mul_mask_ld", g_oin, hh, la, hh, ids, M, hh, g_uvqk, 4 * hh)
        g_la = new(M, hh)
        _c("rt_mul_mask_ld", g_oin, hh, uvqk, 4 * hh, ids, M, hh, g_la, hh)
        g_ad, d_ln2w, d_ln2b = new(M, hh), new(hh), new(hh)
        ws_bytes = lib.rt_layernorm_bwd_workspace_bytes(M, hh)
        ws = torch.empty((max(ws_bytes, 4),), dtype=torch.uint8, device=dev)
        _ln_bwd_split(g_la, attn_d, ln2_w, mean2, rstd2, None, None, 0, 0, M, hh, g_ad, d_ln2w, d_ln2b, ws, ws_bytes, defer)
        if p_attn > 0:
            g_attn = new(M, hh)
            _c("rt_act_dropout_bwd", g_ad, g_ad, ACT_NONE, float(p_attn), seed_a[0], seed_a[1], g_ad.numel(), g_attn)
        else:
            g_attn = g_ad
        has_t = tw is not None
        dtw = None if tw is None else torch.zeros_like(tw)
        dpw = None if pw is None else torch.zeros_like(pw)
        if cu is None:
            _c("rt_hstu_attn_bwd", uvqk[:, 2 * hh:], 4 * hh, uvqk[:, 3 * hh:], 4 * hh, uvqk[:, hh:], 4 * hh, g_attn, hh, ids,
               ts if has_t else None, tw, thr if has_t else None, pw, B, H, L, hd, g_uvqk[:, 2 * hh:], 4 * hh, g_uvqk[:, 3 * hh:], 4 * hh,
               g_uvqk[:, hh:], 4 * hh, dtw, dpw)
        else:
            if rows_real < M:
                g_uvqk[rows_real:, hh:].zero_()               # the kernel writes session rows only; dP sums over every row
            _c("rt_hstu_attn_varlen_bwd", uvqk[:, 2 * hh:], 4 * hh, uvqk[:, 3 * hh:], 4 * hh, uvqk[:, hh:], 4 * hh, g_attn, hh, cu,
               ts if has_t else None, tw, thr if has_t else None, pw, B, H, L, hd, g_uvqk[:, 2 * hh:], 4 * hh, g_uvqk[:, 3 * hh:], 4 * hh,
               g_uvqk[:, hh:], 4 * hh, dtw, dpw)
        g_z = new(M, 4 * hh)
        _c("rt_act_dropout_bwd", g_uvqk, z, ACT_SILU, 0.0, 0, 0, g_uvqk.numel(), g_z)
        d_p = new(d, 4 * hh)
        with _OnSide(dev, defer) as sd:
            side.append(sd)
            sd.uses(n1, g_z, d_p)
            _gemm(n1, d, 0, g_z, 4 * hh, 0, d_p, 4 * hh, None, None, 0, d, 4 * hh, M, 0, sp)       # dP = n^T g_z
        g_n = new(M, d)
        _gemm_w(g_z, 4 * hh, uvqk_p, 4 * hh, 1, ctx.wp[0], g_n, d, None, None, 0, M, d, 4 * hh)   # g_n = g_z P^T
        # ---- n = LN(x0) * m, x0 = x * m (+ the skip connection): one kernel
        g_x, d_ln1w, d_ln1b = new(M, d), new(d), new(d)
        ws_bytes = lib.rt_layernorm_bwd_workspace_bytes(M, d)
        ws = torch.empty((max(ws_bytes, 4),), dtype=torch.uint8, device=dev)
        msk = 1 if cu is None else 0
        ws1_bytes = lib.rt_layernorm_bwd_workspace_bytes(M, d)      # (its own partial sums: the first LayerNorm's combine may still be in flight)
        ws1 = torch.empty((max(ws1_bytes, 4),), dtype=torch.uint8, device=dev)
        _ln_bwd_split(g_n, x0, ln1_w, mean1, rstd1, g_out, ids, msk, msk, M, d, g_x, d_ln1w, d_ln1b, ws1, ws1_bytes, defer)
        if side:
            side[-1].join_now()
        return (g_x, None, None, None, d_ln1w, d_ln1b, d_p, dtw, dpw, d_ln2w, d_ln2b, d_wo, d_bo, None)


def stu_layer(x: torch.Tensor, ids: torch.Tensor, ts: tp.Optional[torch.Tensor], thr: torch.Tensor, B: int, L: int, H: int,
              hd: int, p_attn: float, p_mlp: float, ln_in: tp.Tuple[torch.Tensor, torch.Tensor, float], uvqk_proj: torch.Tensor,
              time_w: tp.Optional[torch.Tensor], pos_w: tp.Optional[torch.Tensor],
              ln_attn: tp.Tuple[torch.Tensor, torch.Tensor, float], out_mlp: tp.Tuple[torch.Tensor, torch.Tensor]) -> torch.Tensor:
    """Fused STU block on [B*L, d] activations (input row mask included); parameters as (weight, bias[, eps])."""
    for t in (x, ln_in[0], uvqk_proj, ln_attn[0], out_mlp[0]):
        _chk(t, "stu_layer")
    return _STULayer.apply(x, ids.reshape(-1), ts, thr, ln_in[0], ln_in[1], uvqk_proj, time_w, pos_w, ln_attn[0], ln_attn[1],
                           out_mlp[0], out_mlp[1], (B, L, H, hd, float(p_attn), float(p_mlp), ln_in[2], ln_attn[2]))


def stu_layer_packed(x: torch.Tensor, cu: torch.Tensor, rows_real: int, ts: tp.Optional[torch.Tensor], thr: torch.Tensor, B: int, window: int,
                     H: int, hd: int, p_attn: float, p_mlp: float, ln_in: tp.Tuple[torch.Tensor, torch.Tensor, float], uvqk_proj: torch.Tensor,
                     time_w: tp.Optional[torch.Tensor], pos_w: tp.Optional[torch.Tensor],
                     ln_attn: tp.Tuple[torch.Tensor, torch.Tensor, float], out_mlp: tp.Tuple[torch.Tensor, torch.Tensor]) -> torch.Tensor:
    """The fused STU block on PACKED rows [Np, d] (session b = rows cu[b] .. cu[b+1]-1, rows_real = cu[B]; ts = the packed timestamps
    of `collate_packed_ts`): `_STULayer` without the masks, on `rt_hstu_attn_varlen_*`."""
    for t in (x, ln_in[0], uvqk_proj, ln_attn[0], out_mlp[0]):
        _chk(t, "stu_layer_packed")
    return _STULayer.apply(x, None, ts, thr, ln_in[0], ln_in[1], uvqk_proj, time_w, pos_w, ln_attn[0], ln_attn[1], out_mlp[0], out_mlp[1],
                           (B, window, H, hd, float(p_attn), float(p_mlp), ln_in[2], ln_attn[2], cu, int(rows_real)))


# --------------------------------------------------------------------------------------------------
# losses
# --------------------------------------------------------------------------------------------------
_PREPARED_PAIRS: tp.Dict[tp.Tuple, torch.Tensor] = {}   # at most ONE entry: the workspace `prepare_sampled_pairs` filled for this step's batch


def _pairs_key(y: torch.Tensor, neg: torch.Tensor, V: int, d: int) -> tp.Tuple:
    return (RNG.step, y.data_ptr(), neg.data_ptr(), int(y.numel()), int(neg.shape[-1]), int(V), int(d))


def prepare_sampled_pairs(y: torch.Tensor, neg: torch.Tensor, V: int, d: int) -> None:
    """The sampled losses' counting sort of the (position, candidate) pairs by candidate id — needed by the backward pass only, a
    function of the ids only — issued on the side stream as soon as the batch exists (`rt_sampled_loss_prepare`): six small launches
    leave the gap between the forward and the backward kernels of a training step, and the training forward writes the pair records
    itself.  NOT called by the stock loop: measured neutral at C2 in round 4 (84.7 vs 84.5 k seqs/s — the 1.7 M rank atomics that hide under the
    forward kernel's gathers cost 83 us as a kernel of their own) and SLOWER in round 6 (81.4 vs 87.6 k: the sort's kernels now sit beside
    the forward pass's three-per-CU attention workgroups); kept as the binding of the entry point, pinned against the in-pass sort
    by tests/test_ops_gpu.py.  `sampled_loss` finds the workspace through (step, the two tensors' storage, sizes); any mismatch (another y / neg, a
    loss that is never called) just leaves it unused."""
    _PREPARED_PAIRS.clear()
    if not (y.is_cuda and neg.is_cuda and y.dtype == torch.int64 and neg.dtype == torch.int64):
        return
    y1 = y.reshape(-1)
    M = int(y1.numel())
    if M == 0 or not y1.is_contiguous() or not neg.is_contiguous() or neg.numel() % M != 0:
        return
    neg2 = neg.reshape(M, -1)
    N = int(neg2.shape[1])
    ws_bytes = _lib.load().rt_sampled_loss_bwd_workspace_bytes(M, N, int(V), int(d))
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=y.device)
    side = _side_fork_forward()
    _c("rt_sampled_loss_prepare", y1, neg2, M, N, int(d), int(V), ws, ws_bytes, stream=side)
    if side is not None:
        _PREP_KEEPALIVE.append((y1, neg2, ws))
    _PREPARED_PAIRS[_pairs_key(y1, neg2, V, d)] = ws


class _SampledLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, sess, table, y, neg, w, loss, cosine, logits_t, beta):
        M, d = sess.shape
        N = neg.shape[-1]
        V = table.shape[0]
        dev = sess.device
        logits = torch.empty((M, N + 1), dtype=torch.float32, device=dev)
        loss_pos = torch.empty((M,), dtype=torch.float32, device=dev)
        out = torch.empty((2,), dtype=torch.float32, device=dev)
        train = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        if train:  # one pass over the candidate rows also yields the unit gradients the backward needs
            ws_bytes = _lib.load().rt_sampled_loss_bwd_workspace_bytes(M, N, V, d)
            ws = _PREPARED_PAIRS.pop(_pairs_key(y, neg, V, d), None)       # sorted ahead of the forward pass (`prepare_sampled_pairs`)?
            prepared = ws is not None and ws.numel() >= ws_bytes
            if prepared:
                if _PREP_KEEPALIVE:
                    join_side_streams()                                     # (the sort ran on the side stream)
            else:
                ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
            du = torch.empty((M, d), dtype=torch.float32, device=dev)
            _c("rt_sampled_loss_fwd_train", sess, sess.stride(0), table, y, neg, w, M, N, d, V, loss, int(cosine),
               float(logits_t), float(beta), logits, loss_pos, du, d, ws, ws_bytes, 1 if prepared else 0)
            ctx.prepared = prepared
            ctx.save_for_backward(sess, table, y, neg, logits, out, du, ws)
        else:
            _c("rt_sampled_loss_fwd", sess, sess.stride(0), table, y, neg, w, M, N, d, loss, int(cosine), float(logits_t),
               float(beta), logits, loss_pos)
        _c("rt_loss_reduce", loss_pos, y, M, 0 if loss == LOSS_SAMPLED_SOFTMAX else 1, out)
        ctx.meta = (cosine, logits_t)
        ctx.mark_non_differentiable(logits)
        ctx.set_materialize_grads(False)     # no [M, 1 + N] zeros for the logits' unused gradient at the start of every backward pass
        return out[0], logits

    @staticmethod
    def backward(ctx, gloss, _glogits):
        sess, table, y, neg, logits, out, du, ws = ctx.saved_tensors
        cosine, logits_t = ctx.meta
        M, d = sess.shape
        N = neg.shape[-1]
        V = table.shape[0]
        if gloss is None:                    # (set_materialize_grads(False): nothing asked for the loss)
            return (None,) * 9
        d_sess = torch.empty((M, d), dtype=torch.float32, device=sess.device)
        d_table = _new_table_grad(table)     # (every row is written: zeros where no candidate fell)
        # The upstream gradient stays on the device: the kernels scale by upstream[0] / norm[0] themselves — no device -> host copy (it made
        # the host wait for the whole forward pass at the start of every backward pass), and no divide launch between the loss and the
        # first backward kernel either (round 6)
        up = gloss.reshape(1) if gloss.dtype == torch.float32 else gloss.reshape(1).to(torch.float32)
        args = (sess, sess.stride(0), table, y, neg, M, N, d, V, int(cosine), float(logits_t), logits, out[1:], 1.0, up, du, d)
        prep = 1 if getattr(ctx, "prepared", False) else 0
        # rt_sampled_loss_bwd accepts either output as NULL: the session half feeds the layer backward (main stream), the table half —
        # read by the optimiser only — goes to the side stream when autograd will merely adopt its result AND an embedding lookup of
        # this forward pass will pick it up as its sink (see `_TABLE_SINK_EXPECTED`); otherwise one call on the main stream.
        side = _native_side_fork() if (_LOSS_TABLE_ON_SIDE and ctx.needs_input_grad[1] and _steals_grad(table)
                                       and _table_sink_expected(table)) else None
        if side is None:
            _c("rt_sampled_loss_bwd", *args, d_sess, d, d_table, ws, ws.numel(), prep)
        else:
            # (round 2 measured this slower, round 3 equal; with the chain kernels of round 4 it is +4 % on the C2 step: `_LOSS_TABLE_ON_SIDE`)
            _c("rt_sampled_loss_bwd", *args, d_sess, d, None, ws, ws.numel(), prep)
            _c("rt_sampled_loss_bwd", *args, None, d, d_table, ws, ws.numel(), prep, stream=side)
            _lib.check(_lib.load().rt_side_mark(), "rt_side_mark")      # the embedding backward waits for THIS point, not for the weight gradients behind it
            # NOT d_table: an extra reference would make autograd's AccumulateGrad clone it (on the main stream, before the side stream
            # has written it) instead of adopting it; as `table.grad` it outlives the join anyway
            _NATIVE_KEEPALIVE.append((sess, table, y, neg, logits, out, du, ws, up))
            _TABLE_GRAD_ON_SIDE.add(d_table.data_ptr())
        if ctx.needs_input_grad[1]:   # only a gradient autograd will hand to the table can serve as the embedding node's sink
            _offer_table_grad(table, d_table)
        return d_sess, d_table, None, None, None, None, None, None, None


def sampled_loss(sess: torch.Tensor, table: torch.Tensor, y: torch.Tensor, neg: torch.Tensor, w: torch.Tensor, loss: int,
                 cosine: bool, logits_t: float, gbce_beta: float = 0.0) -> tp.Tuple[torch.Tensor, torch.Tensor]:
    """-> (scalar loss, logits [M, 1+N] / logits_t).  y/w [M], neg [M,N]; positions with y == 0 are ignored."""
    M = sess.shape[0]
    return _SampledLoss.apply(sess, table, y.reshape(-1), neg.reshape(M, -1), w.reshape(-1).contiguous(), loss, cosine,
                              logits_t, gbce_beta)


def _padded_table(table: torch.Tensor) -> tp.Optional[torch.Tensor]:
    """[Vp, d] view over a table whose owner (lightning.FlatAdam) keeps zero rows behind it up to a multiple of 128, or None."""
    rows = getattr(table, "_rt_rows_padded", None)
    if rows is None or not table.is_contiguous() or rows < table.shape[0] or rows % 128 != 0 or table.shape[1] % 32 != 0:
        return None
    need = (table.storage_offset() + rows * table.shape[1]) * table.element_size()
    if table.untyped_storage().nbytes() < need:
        return None
    return torch.as_strided(table.detach(), (rows, table.shape[1]), (table.shape[1], 1), table.storage_offset())


class _SoftmaxLoss(torch.autograd.Function):
    """Full-catalog softmax CE over the active positions (lightning.py:145-162).

    The three products (logits = S E^T, dS = G E, dE = G^T S) run over R active rows and V catalog rows — neither a
    multiple of the 128-wide GEMM tile in general, which would send 114 GFLOP per C3 step through the ragged-edge kernel.
    When the table owns zero rows up to Vp (a multiple of 128, see `_padded_table`) the active rows are padded with zero
    rows to Rp as well: every product is then an exact tile grid (LDS-DMA GEMM), the pad rows / columns of the logits are
    exact zeros (zero operand rows) and never enter the softmax (`rt_softmax_ce_rows` walks R x V)."""

    @staticmethod
    def forward(ctx, sess, table, act_idx, y_act, w_act, logits_t, M_total):
        if _PREP_KEEPALIVE:          # the embedding's sort (side stream, issued at the start of the forward pass) is long done: joining it
            join_side_streams()      # HERE costs nothing, joining it in the embedding backward would wait for the weight gradients
        R = act_idx.numel()
        V, d = table.shape
        tp_ = _padded_table(table)
        Vp = V if tp_ is None else tp_.shape[0]
        Rp = R if tp_ is None else (R + 127) // 128 * 128
        tab = table if tp_ is None else tp_
        s_act = (torch.empty if Rp == R else torch.zeros)((Rp, d), dtype=torch.float32, device=sess.device)
        _c("rt_gather_rows", sess, sess.stride(0), act_idx, R, d, s_act, d)
        logits = torch.empty((Rp, Vp), dtype=torch.float32, device=sess.device)
        _gemm(s_act, d, 1, tab, tab.stride(0), 1, logits, Vp, None, None, 0, Rp, Vp, d)
        loss_pos = torch.empty((R,), dtype=torch.float32, device=sess.device)
        lse = torch.empty((R,), dtype=torch.float32, device=sess.device)
        out = torch.empty((2,), dtype=torch.float32, device=sess.device)
        _c("rt_softmax_ce_rows", logits, Vp, R, V, y_act, w_act, float(logits_t), 0, None, 1.0, None, loss_pos, lse)
        _c("rt_loss_reduce", loss_pos, y_act, R, 0, out)
        ctx.save_for_backward(s_act, tab, act_idx, y_act, w_act, logits, lse, out)
        ctx.meta = (logits_t, M_total, sess.shape[1], R, V)
        return out[0]

    @staticmethod
    def backward(ctx, gloss):
        s_act, tab, act_idx, y_act, w_act, logits, lse, out = ctx.saved_tensors
        logits_t, M_total, d, R, V = ctx.meta
        Rp, Vp = logits.shape
        # logits := (softmax - onehot) * w * g / (norm * t), in place (the buffer is ours); pad rows / columns stay zero
        up = gloss.reshape(1) if gloss.dtype == torch.float32 else gloss.reshape(1).to(torch.float32)    # stays on the device (see _SampledLoss.backward)
        _c("rt_softmax_ce_rows", logits, Vp, R, V, y_act, w_act, float(logits_t), 1, out[1:], 1.0, up, None, lse)
        ds_act = torch.empty((Rp, d), dtype=torch.float32, device=logits.device)
        _gemm(logits, Vp, 1, tab, tab.stride(0), 0, ds_act, d, None, None, 0, Rp, d, Vp, 0, _deep_k_splits(Rp, d, Vp))  # dS = G @ E
        d_tab = torch.empty((Vp, d), dtype=torch.float32, device=logits.device)
        _gemm(logits, Vp, 0, s_act, d, 0, d_tab, d, None, None, 0, Vp, d, Rp, 0, _wgrad_splits(Rp, Vp, d))  # dE = G^T @ S
        d_table = d_tab[:V]
        d_table[0].zero_()  # padding_idx row never receives gradient (item_net.py:260-264)
        if ctx.needs_input_grad[1]:
            _offer_table_grad(tab[:V], d_table)
        d_sess = torch.zeros((M_total, d), dtype=torch.float32, device=logits.device)
        _c("rt_scatter_rows", ds_act, d, act_idx, R, d, d_sess, d)
        return d_sess, d_table, None, None, None, None, None


# ---- packed rows <-> padded window (for stacks whose attention kernels work on the [B, L] window) ------------------------
def padded_index(cu: torch.Tensor, B: int, window: int, n_rows: int) -> tp.Tuple[torch.Tensor, torch.Tensor]:
    """Where every packed row sits in the left-padded [B, window] layout: -> (idx [n_rows] int64, ids [B * window + 1] int64).
    Session b = packed rows cu[b] .. cu[b+1] - 1 occupies the LAST cu[b+1] - cu[b] slots of window b; rows behind cu[B] (the unused
    tail of the row block) go to the extra slot B * window, which no real query reads.  ids: 1 at occupied slots, 0 at pad slots —
    what the window kernels derive their masks from.  Device ops only (no host round trip)."""
    dev = cu.device
    r = torch.arange(n_rows, dtype=torch.int64, device=dev)
    b = torch.searchsorted(cu[1:B + 1].contiguous(), r, right=True)                 # session of every row; B for the tail rows
    bc = b.clamp(max=B - 1)
    idx = bc * window + (window - (cu[bc + 1] - cu[bc])) + (r - cu[bc])
    idx = torch.where(b >= B, torch.full_like(idx, B * window), idx)
    ids = torch.zeros((B * window + 1,), dtype=torch.int64, device=dev)
    ids[idx] = 1
    ids[B * window] = 0
    return idx, ids


class _ScatterRows(torch.autograd.Function):
    """out [n_out, d] = 0; out[idx[r]] = src[r] (idx unique except for rows sent to a dump slot).  Backward: gather."""

    @staticmethod
    def forward(ctx, src, idx, n_out):
        src = src.contiguous()
        R, d = src.shape
        out = torch.zeros((n_out, d), dtype=torch.float32, device=src.device)
        _c("rt_scatter_rows", src, d, idx, R, d, out, d)
        ctx.save_for_backward(idx)
        return out

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        g = g.contiguous()
        R, d = idx.numel(), g.shape[1]
        out = torch.empty((R, d), dtype=torch.float32, device=g.device)
        _c("rt_gather_rows", g, d, idx, R, d, out, d)
        return out, None, None


class _GatherRows(torch.autograd.Function):
    """out[r] = src[idx[r]].  Backward: scatter into zeros (rows sharing a dump slot collide there: that slot's gradient is unused)."""

    @staticmethod
    def forward(ctx, src, idx):
        src = src.contiguous()
        R, d = idx.numel(), src.shape[1]
        out = torch.empty((R, d), dtype=torch.float32, device=src.device)
        _c("rt_gather_rows", src, d, idx, R, d, out, d)
        ctx.save_for_backward(idx)
        ctx.n_src = src.shape[0]
        return out

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        g = g.contiguous()
        out = torch.zeros((ctx.n_src, g.shape[1]), dtype=torch.float32, device=g.device)
        _c("rt_scatter_rows", g, g.shape[1], idx, idx.numel(), g.shape[1], out, g.shape[1])
        return out, None


def scatter_rows(src: torch.Tensor, idx: torch.Tensor, n_out: int) -> torch.Tensor:
    return _ScatterRows.apply(_chk(src, "scatter_rows"), idx, n_out)


def gather_rows(src: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    return _GatherRows.apply(_chk(src, "gather_rows"), idx)


def softmax_loss(sess: torch.Tensor, table: torch.Tensor, act_idx: torch.Tensor, y_act: torch.Tensor, w_act: torch.Tensor,
                 logits_t: float) -> torch.Tensor:
    """sess [M,d] (already L2-normalised for cosine), table [V,d]; act_idx = positions with y != 0 (int64, ascending)."""
    return _SoftmaxLoss.apply(sess, table, act_idx, y_act, w_act, logits_t, sess.shape[0])
