"""`HipRanker` — drop-in for the reference's `TorchRanker` (rectools/models/rank/rank_torch.py:30-223).

Same constructor arguments, same `rank(subject_ids, k, filter_pairs_csr, sorted_object_whitelist)`
contract (`Ranker` protocol, rectools/models/rank/rank.py:33-64), same return triplet (flat arrays grouped
by subject, best first) and the same `ValueError` for a CSR/subject row mismatch (rank_torch.py:107-109).

What differs is *how*: factors stay on the GPU, the viewed-items filter stays in CSR form, and one
HIP launch sequence (`rt_topk_score`, csrc/rt_topk.hip) streams the catalog once per user batch with
scores living only in MFMA accumulators — there is no `[batch, n_items]` score matrix, no `toarray()`,
no per-batch H2D/D2H.  There is no CPU fallback: without the HIP library / a GPU this raises.
"""
from __future__ import annotations

import os
import typing as tp
from enum import Enum

import numpy as np
import torch
from scipy import sparse

from . import _lib


class Distance(str, Enum):
    """Distance metric (mirror of rectools/models/rank/rank.py:25-30)."""

    DOT = "dot"  # Bigger value means closer vectors
    COSINE = "cosine"  # Bigger value means closer vectors
    EUCLIDEAN = "euclidean"  # Smaller value means closer vectors


_DIST_CODE = {Distance.DOT: 0, Distance.COSINE: 1, Distance.EUCLIDEAN: 2}


class DeviceCSR:
    """Viewed-pairs filter kept on the GPU in CSR form (int64 indptr, int32 ascending indices).

    The reference densifies the filter per 128-user batch on the host (`toarray()`, rank_torch.py:141);
    here it is uploaded once and binary-searched in-kernel only for candidates that beat the threshold.
    """

    def __init__(self, indptr: torch.Tensor, indices: torch.Tensor, shape: tp.Tuple[int, int]):
        self.indptr, self.indices, self.shape = indptr, indices, shape
        self._hash: tp.Optional[torch.Tensor] = None

    def hash_tables(self) -> torch.Tensor:
        """Per-user hash sets over the rows (built once on the device): the selection slow path of `rt_topk_score`
        tests membership in 1-2 loads instead of a binary search of the CSR row."""
        if self._hash is None:
            lib = _lib.load()
            n_users = self.shape[0]
            nnz = int(self.indptr[-1].item()) if n_users else 0
            h = torch.empty((lib.rt_filter_hash_bytes(n_users, nnz) // 4,), dtype=torch.int32, device=self.indptr.device)
            with torch.cuda.device(self.indptr.device):
                _lib.check(lib.rt_filter_hash_build(_lib.ptr(self.indptr), _lib.ptr(self.indices), n_users, nnz, _lib.ptr(h),
                                                    _lib.current_stream()), "rt_filter_hash_build")
            self._hash = h
        return self._hash

    @classmethod
    def from_scipy(cls, csr: sparse.csr_matrix, device: tp.Union[torch.device, str]) -> "DeviceCSR":
        csr = sparse.csr_matrix(csr, copy=True)
        csr.sum_duplicates()
        csr.eliminate_zeros()  # reference masks on `toarray() != 0` (rank_torch.py:141-143)
        csr.sort_indices()
        indptr = torch.from_numpy(csr.indptr.astype(np.int64)).to(device)
        indices = torch.from_numpy(csr.indices.astype(np.int32)).to(device)
        if indices.numel() == 0:
            indices = torch.zeros((1,), dtype=torch.int32, device=device)
        return cls(indptr, indices, (int(csr.shape[0]), int(csr.shape[1])))


def _pad4(t: torch.Tensor) -> torch.Tensor:
    """fp32 contiguous [n, d] with d padded to a multiple of 4 floats (16-byte rows for float4 loads)."""
    d = t.shape[1]
    if d % 4 != 0:
        t = torch.nn.functional.pad(t, (0, 4 - d % 4))
    return t.contiguous()


class HipRanker:
    """
    Ranker based on the gfx950 top-k scoring kernel.

    Parameters
    ----------
    distance : Distance
        Distance metric.
    device : torch.device | str
        HIP device to calculate on (``"cuda"`` / ``"cuda:0"`` on ROCm).
    subjects_factors : np.ndarray | sparse.csr_matrix | torch.Tensor
        Subjects embeddings, shape (n_subjects, n_factors).
    objects_factors : np.ndarray | torch.Tensor
        Objects embeddings, shape (n_objects, n_factors).
    batch_size : int, default 128
        Kept for interface compatibility with `TorchRanker`; here it is the number of users that share one
        pass over the catalog (32, 64 or 128; other values are rounded).  ``None`` -> library default.
    dtype : torch.dtype, optional, default ``torch.float32``
        Only float32 is supported (the reference's default).
    """

    def __init__(
        self,
        distance: Distance,
        device: tp.Union[torch.device, str],
        subjects_factors: tp.Union[np.ndarray, sparse.csr_matrix, torch.Tensor],
        objects_factors: tp.Union[np.ndarray, torch.Tensor],
        batch_size: tp.Optional[int] = None,
        dtype: tp.Optional[torch.dtype] = torch.float32,
        two_stage: tp.Optional[bool] = None,
    ):
        if dtype not in (None, torch.float32):
            raise NotImplementedError("HipRanker computes in float32 only")
        self.distance = Distance(distance)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.HipLibraryError("HipRanker needs a HIP device (no CPU fallback); got device=%s" % device)
        _lib.require_gpu()
        self._lib = _lib.load()
        self.batch_size = batch_size
        self.dtype = torch.float32
        self._higher_is_better = self.distance != Distance.EUCLIDEAN
        self.subjects_factors = self._to_device(subjects_factors)
        self.objects_factors = self._to_device(objects_factors)
        if self.subjects_factors.shape[1] != self.objects_factors.shape[1]:
            raise ValueError("subjects and objects factors must have the same number of columns")
        self._workspace: tp.Optional[torch.Tensor] = None
        # two-stage top-k (include/rectools_hip.h K12c): hm-image coarse pass + exact pass over the candidates, with a per-user proof
        # that the result is the single-stage kernel's.  None = where it pays (dot products, many users per call); RT_TOPK_TWO_STAGE=0/1
        # forces it off / on wherever it applies
        env = os.environ.get("RT_TOPK_TWO_STAGE", "auto")
        self.two_stage: tp.Optional[bool] = (None if env == "auto" else env == "1") if two_stage is None else bool(two_stage)
        self._items_hm: tp.Optional[torch.Tensor] = None      # hm image of objects_factors (rt_to_hm_rows), built on first use
        self._items_h: tp.Optional[torch.Tensor] = None       # h-only (one bf16 per value) image: the HBM-bound regime
        self._items_frag: tp.Optional[torch.Tensor] = None    # ... in fragment-major form (RT_TOPK_FRAG=1)
        self._h_only_off = os.environ.get("RT_TOPK_H_ONLY", "1") == "0"
        self._h_only_strikes = 0
        self._max_item_norm = 0.0
        self.two_stage_stats = {"calls": 0, "fallbacks": 0, "unproven_users": 0, "h_only_calls": 0}
        self._unsettled: tp.List[tp.Tuple[torch.Tensor, bool, tp.Tuple]] = []      # two-stage calls whose proof flags nobody has read yet (`settle`)

    # ---- catalog images outlive a ranker -----------------------------------------------------------------
    _IMAGE_FIELDS = ("_items_hm", "_items_h", "_items_frag", "_h_only_off", "_h_only_strikes", "_max_item_norm")

    def export_images(self) -> tp.Dict[str, tp.Any]:
        """The coarse-pass images of `objects_factors` built so far (+ what the ranker learnt about the catalog: the largest item norm,
        whether the one-plane bound holds on it).  `recommend()` builds a ranker per call (new user factors every time) against a
        catalog that only changes when the model trains: the model keeps this dict next to a version key of its item embeddings and
        hands it to the next ranker (`adopt_images`) — no catalog pass, no device -> host read of the norm per call."""
        return {k: getattr(self, k) for k in self._IMAGE_FIELDS}

    def adopt_images(self, images: tp.Optional[tp.Dict[str, tp.Any]]) -> None:
        """Take over images exported by a ranker over the SAME objects_factors (the caller vouches for that)."""
        if not images:
            return
        n, d = self.objects_factors.shape
        for k in ("_items_hm", "_items_h"):
            t = images.get(k)
            if t is not None and (t.shape[0] != n or t.device != self.objects_factors.device):
                return      # not this catalog: keep nothing
        for k in self._IMAGE_FIELDS:
            if k in images:
                setattr(self, k, images[k])

    def _to_device(self, tensor: tp.Union[np.ndarray, sparse.csr_matrix, torch.Tensor]) -> torch.Tensor:
        # mirrors TorchRanker._normalize_tensor (rank_torch.py:210-223), then moves to the device once
        if isinstance(tensor, sparse.csr_matrix):
            tensor = tensor.toarray()
        if isinstance(tensor, np.ndarray):
            tensor = torch.from_numpy(np.ascontiguousarray(tensor))
        tensor = tensor.detach().to(device=self.device, dtype=torch.float32)
        if tensor.dim() != 2:
            raise ValueError("factors must be 2-dimensional")
        return _pad4(tensor)

    # ---- two-stage top-k ---------------------------------------------------------------------------------
    CANDIDATES = 64           # k_cand: coarse candidates handed to the exact pass per user
    TWO_STAGE_MIN_USERS = 17   # up to 16 users the 16-user tile of the single-stage kernel streams the catalog at the HBM roofline; from 17 up
    #                            the matrix pipe binds and the coarse pass wins (5 M x 512: 32 users 2.36 vs 2.77 ms, 64 users 2.60 vs 4.08 ms)

    FRAG_MIN_ITEMS = 500_000  # catalogs from which the fragment-major coarse pass is the default for >= 128 users per call
    RUNS_PER_USERS = 150      # one-plane pass: more than n_users / 150 separate runs of unproven users cost more than the (h, m) pass again

    def _two_stage_applies(self, kk: int, n_cand: int, n_subj: int) -> bool:
        d = self.objects_factors.shape[1]
        if self.two_stage is False or self.distance not in (Distance.DOT, Distance.COSINE) or d % 32 != 0 or d > 2048 or kk > 16 \
                or n_cand < 8 * self.CANDIDATES:
            return False
        return True if self.two_stage else (n_subj >= int(os.environ.get("RT_TOPK_TWO_STAGE_MIN_USERS", self.TWO_STAGE_MIN_USERS))
                                            or self._h_only_applies(n_subj))

    H_ONLY_MIN_BYTES = 256 << 20   # up to 16 users the one-plane pass only pays when the catalog does not live in L2 / Infinity Cache

    def _h_only_applies(self, n_subj: int) -> bool:
        """The one-plane coarse pass (ONE bf16 per value: half the bytes of the (h, m) image and a quarter of its matrix-pipe work, coarse
        error 2^-7 |u| |v|).  A few users against a big catalog: the pass is bound by the bytes it streams (5 M x 512, 16 users: 1.16 vs
        1.77 ms for the fp32 rows).  Many users: the (h, m) pass is bound by its four bf16 products per fp32 product (5 M x 512, 4096
        users: 58.6 vs 101.6 ms; 26,744 x 256, 16,384 users: 2.47 vs 3.08 ms).  Switched off for a ranker whose catalog defeated the
        wider error bound (near-duplicate items): the (h, m) image serves it from then on."""
        O = self.objects_factors
        if self._h_only_off or O.shape[1] % 64 != 0 or O.stride(0) != O.shape[1]:
            return False
        if n_subj >= int(os.environ.get("RT_TOPK_TWO_STAGE_MIN_USERS", self.TWO_STAGE_MIN_USERS)):
            return True
        return O.numel() * 4 >= int(os.environ.get("RT_TOPK_H_ONLY_MIN_BYTES", self.H_ONLY_MIN_BYTES))

    def _fragments(self, img: torch.Tensor, n_rows: int, d: int) -> torch.Tensor:
        """Fragment-major form of a one-plane image (`rt_one_plane_to_fragments`), rows padded to a multiple of 128 with zeros."""
        rows_pad = (n_rows + 127) // 128 * 128
        out = torch.empty((rows_pad, d // 2), dtype=torch.int32, device=self.device)
        status = self._lib.rt_one_plane_to_fragments(_lib.ptr(img), img.stride(0), n_rows, d, _lib.ptr(out), rows_pad, _lib.current_stream())
        _lib.check(status, "rt_one_plane_to_fragments")
        return out

    def _hm_image(self, src: torch.Tensor, rows: tp.Optional[torch.Tensor], n_rows: int, h_only: bool = False
                  ) -> tp.Tuple[torch.Tensor, torch.Tensor]:
        d = src.shape[1]
        w = d // 2 if h_only else d
        img = torch.empty((n_rows, w), dtype=torch.int32, device=self.device)
        norms = torch.empty((n_rows,), dtype=torch.float32, device=self.device)
        mode = (1 if self.distance == Distance.COSINE else 0) | (2 if h_only else 0)
        status = self._lib.rt_to_hm_rows(_lib.ptr(src), src.stride(0), _lib.ptr(rows), n_rows, d, mode,
                                         _lib.ptr(img), w, _lib.ptr(norms), _lib.current_stream())
        _lib.check(status, "rt_to_hm_rows")
        return img, norms

    def _rank_exact(self, ids_t, scores_t, counts_t, rows_t, u0, n, whitelist_t, n_cand, id_offset, kk, indptr_t, indices_t, hash_t,
                    upp) -> None:
        """`rt_topk_score` for the users [u0, u0 + n) of the call, written into their rows of the outputs."""
        S, O = self.subjects_factors, self.objects_factors
        ws_bytes = self._lib.rt_topk_workspace_bytes(n, n_cand, kk, upp)
        if self._workspace is None or self._workspace.numel() < ws_bytes:
            self._workspace = torch.empty((ws_bytes,), dtype=torch.uint8, device=self.device)
        off = lambda t, elems, size: None if t is None else t.data_ptr() + elems * size   # noqa: E731
        with torch.cuda.device(self.device):
            status = self._lib.rt_topk_score(
                S.data_ptr() + (0 if rows_t is not None else 4 * u0 * S.stride(0)), S.stride(0), off(rows_t, u0, 8), n,
                O.data_ptr() + 4 * id_offset * O.stride(0), O.stride(0), _lib.ptr(whitelist_t), n_cand, id_offset, O.shape[1],
                _DIST_CODE[self.distance], kk,
                # the filter of user u0 + i: indptr entry u0 + i; its hash table sits 4 (indptr[u] + u) ints into the hash array
                off(indptr_t, u0, 8), _lib.ptr(indices_t), off(hash_t, 4 * u0, 4),
                ids_t.data_ptr() + 8 * u0 * kk, scores_t.data_ptr() + 4 * u0 * kk, counts_t.data_ptr() + 4 * u0,
                _lib.ptr(self._workspace), self._workspace.numel(), upp, _lib.current_stream())
        _lib.check(status, "rt_topk_score")

    def _rank_two_stage(self, ids_t, scores_t, counts_t, rows_t, n_subj, whitelist_t, n_cand, id_offset, kk, indptr_t, indices_t,
                        hash_t, upp, settle: bool = True) -> None:
        """Coarse pass over the hm images + exact pass over CANDIDATES candidates per user (`rt_topk_score_two_stage`); users whose
        result the kernel could not prove complete are ranked again by the single-stage kernel (one device -> host read of the
        flags: the only synchronisation of the call — `settle=False` leaves it to `settle()`)."""
        S, O = self.subjects_factors, self.objects_factors
        d, kc, dev = O.shape[1], self.CANDIDATES, self.device
        self.two_stage_stats["calls"] += 1
        # users per catalog pass: 128 (lists in global memory, half the ring traffic per flop) pays on catalogs long enough that the
        # selection slow path is rare; small catalogs keep the 64-user tile with its LDS lists
        h_only = self._h_only_applies(n_subj)
        # (the one-plane pass re-reads the catalog image once per user tile and is bound by that stream: 64-user tiles with their LDS lists)
        upp2 = upp if upp > 0 else (128 if n_cand >= 500_000 and not h_only else 64)
        with torch.cuda.device(dev):
            if h_only:
                self.two_stage_stats["h_only_calls"] += 1
                if n_subj <= 32:
                    upp2 = 32
                if self._items_h is None:
                    self._items_h, item_norms = self._hm_image(O, None, O.shape[0], h_only=True)
                    self._max_item_norm = float(item_norms.max())
                items_img, img_row_bytes = self._items_h, 2 * d
            else:
                if self._items_hm is None:
                    self._items_hm, item_norms = self._hm_image(O, None, O.shape[0])
                    self._max_item_norm = float(item_norms.max())
                items_img, img_row_bytes = self._items_hm, 4 * d
            users_hm, user_norms = self._hm_image(S, rows_t, n_subj, h_only=h_only)
            # fragment-major images: item fragments straight into the matrix operand, the user tile resident in LDS (round 4, visit v4a:
            # 5 M x 512, 4,096 users 58.7 -> 42.0 ms, ids / counts / score bits unchanged).  Default from FRAG_MIN_ITEMS items up — the
            # second image costs another 2 d bytes per item and one more catalog pass to build; RT_TOPK_FRAG=0 / 1 forces it off / on
            frag_env = os.environ.get("RT_TOPK_FRAG", "auto")
            frag = (h_only and (frag_env == "1" or (frag_env == "auto" and O.shape[0] >= self.FRAG_MIN_ITEMS)) and whitelist_t is None
                    and d % 128 == 0 and id_offset % 128 == 0 and n_subj >= 128 and 128 * d * 2 <= 144 * 1024)
            if frag:
                if self._items_frag is None:
                    self._items_frag = self._fragments(self._items_h, O.shape[0], d)
                items_img, upp2 = self._items_frag, 128
                users_hm = self._fragments(users_hm, n_subj, d)
            unproven = torch.empty((n_subj,), dtype=torch.int32, device=dev)
            ws_bytes = self._lib.rt_topk_two_stage_workspace_bytes(n_subj, n_cand, kk, kc, upp2)
            if self._workspace is None or self._workspace.numel() < ws_bytes:
                self._workspace = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
            status = self._lib.rt_topk_score_two_stage(
                _lib.ptr(S), S.stride(0), _lib.ptr(rows_t), n_subj, O.data_ptr() + 4 * id_offset * O.stride(0), O.stride(0),
                _lib.ptr(users_hm), items_img.data_ptr() + img_row_bytes * id_offset, (2 if frag else 1) if h_only else 0, _lib.ptr(user_norms),
                self._max_item_norm,
                _lib.ptr(whitelist_t), n_cand, id_offset, d, _DIST_CODE[self.distance], kk, kc, _lib.ptr(indptr_t), _lib.ptr(indices_t),
                _lib.ptr(hash_t),
                _lib.ptr(ids_t), _lib.ptr(scores_t), _lib.ptr(counts_t), _lib.ptr(unproven), _lib.ptr(self._workspace),
                self._workspace.numel(), upp2, _lib.current_stream())
            _lib.check(status, "rt_topk_score_two_stage")
        args = (ids_t, scores_t, counts_t, rows_t, n_subj, whitelist_t, n_cand, id_offset, kk, indptr_t, indices_t, hash_t, upp)
        if not settle:      # the caller reads the flags later, for all its calls at once
            self._unsettled.append((unproven, h_only, args))
            if len(self._unsettled) >= 4096:      # (a caller that never settles must not pin every call's tensors for ever)
                self.settle()
            return
        self._repair(np.flatnonzero(unproven.cpu().numpy()), h_only, args)

    def settle(self) -> int:
        """Finish every `rank_device(..., settle=False)` call issued so far: ONE device -> host read of all their proof flags, then
        the users a coarse pass could not prove are ranked again into the tensors those calls returned.  Until then such a call's
        results are the best k CANDIDATES (complete for every proven user — in practice all: `two_stage_stats`).  -> users repaired."""
        pending, self._unsettled = self._unsettled, []
        if not pending:
            return 0
        with torch.cuda.device(self.device):
            flags = torch.cat([u for u, _, _ in pending]).cpu().numpy()
        repaired, at = 0, 0
        for unproven, h_only, args in pending:
            bad = np.flatnonzero(flags[at:at + unproven.numel()])
            at += unproven.numel()
            repaired += len(bad)
            self._repair(bad, h_only, args)
        return repaired

    def _repair(self, bad: np.ndarray, h_only: bool, args: tp.Tuple) -> None:
        """Rank the users `bad` of a two-stage call again (see `_rank_two_stage`); `args`: that call's arguments."""
        ids_t, scores_t, counts_t, rows_t, n_subj, whitelist_t, n_cand, id_offset, kk, indptr_t, indices_t, hash_t, upp = args
        if len(bad) == 0:
            return
        self.two_stage_stats["unproven_users"] += int(len(bad))
        starts = bad[np.r_[True, np.diff(bad) > 1]]
        ends = bad[np.r_[np.diff(bad) > 1, True]] + 1
        wide = n_subj >= int(os.environ.get("RT_TOPK_TWO_STAGE_MIN_USERS", self.TWO_STAGE_MIN_USERS)) or self.two_stage is True
        exact_upp = upp if upp > 16 else 32      # the 32-wide engine: the arithmetic the exact pass mirrors
        if h_only and wide and (8 * len(bad) > n_subj or len(starts) > max(2, n_subj // self.RUNS_PER_USERS)):
            # the one-plane bound is too wide for this catalog (or leaves more single runs — a catalog pass each — than a second coarse
            # pass costs): the whole call again over the (h, m) image, whose bound is 2^-7 of this one
            self.two_stage_stats["fallbacks"] += 1
            self._h_only_strikes += 1
            if 8 * len(bad) > n_subj or self._h_only_strikes >= 2:
                self._h_only_off = True
            saved, self._h_only_off = self._h_only_off, True
            try:
                self._rank_two_stage(ids_t, scores_t, counts_t, rows_t, n_subj, whitelist_t, n_cand, id_offset, kk, indptr_t, indices_t,
                                     hash_t, upp)
            finally:
                self._h_only_off = saved
            return
        if 8 * len(bad) > n_subj:       # the catalog defeats the coarse pass (near-duplicates): the whole call on the exact kernel
            self.two_stage_stats["fallbacks"] += 1
            if h_only:                  # ... and the one-plane image is not tried again on this catalog
                self._h_only_off = True
            self._rank_exact(ids_t, scores_t, counts_t, rows_t, 0, n_subj, whitelist_t, n_cand, id_offset, kk, indptr_t, indices_t,
                             hash_t, exact_upp)
            return
        for u0, u1 in zip(starts, ends):    # runs of consecutive unproven users, on the 32-wide engine
            self._rank_exact(ids_t, scores_t, counts_t, rows_t, int(u0), int(u1 - u0), whitelist_t, n_cand, id_offset, kk, indptr_t,
                             indices_t, hash_t, exact_upp)

    def rank(
        self,
        subject_ids: tp.Sequence[int],
        k: tp.Optional[int] = None,
        filter_pairs_csr: tp.Optional[sparse.csr_matrix] = None,
        sorted_object_whitelist: tp.Optional[np.ndarray] = None,
    ) -> tp.Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """Rank objects for every subject; see the `Ranker` protocol (rank.py:33-64 of the reference)."""
        ids_t, scores_t, counts_t, subject_ids = self.rank_device(
            subject_ids, k, filter_pairs_csr, sorted_object_whitelist
        )
        n_subj, kk = ids_t.shape
        counts = counts_t.cpu().numpy()
        if filter_pairs_csr is not None and self.distance == Distance.EUCLIDEAN and kk > 0:
            # Reference quirk kept for drop-in parity: the filter writes -inf, but EUCLIDEAN takes the SMALLEST
            # k, so filtered pairs win the top-k and are then dropped (rank_torch.py:138-152,167-171): a user
            # with f filtered candidates gets only the best k - f unfiltered items.
            csr = sparse.csr_matrix(filter_pairs_csr) if not isinstance(filter_pairs_csr, DeviceCSR) else None
            if csr is None:
                raise NotImplementedError("EUCLIDEAN ranking with a DeviceCSR filter is not supported")
            cols = np.arange(csr.shape[1]) if sorted_object_whitelist is None else np.asarray(sorted_object_whitelist)
            n_filtered = np.asarray((csr[:, cols] != 0).sum(axis=1)).reshape(-1)
            counts = np.minimum(counts, np.maximum(kk - n_filtered, 0))
        ids = ids_t.cpu().numpy()
        scores = scores_t.cpu().numpy()
        valid = np.arange(kk)[None, :] < counts[:, None]
        all_target_ids = np.repeat(subject_ids, kk).reshape(n_subj, kk)[valid]
        all_reco_ids = ids[valid]
        all_scores = scores[valid]
        if filter_pairs_csr is not None:
            # reference drops -inf rows only when a filter was given (rank_torch.py:167-171)
            keep = all_scores > -np.inf
            all_target_ids, all_reco_ids, all_scores = all_target_ids[keep], all_reco_ids[keep], all_scores[keep]
        return all_target_ids, all_reco_ids, all_scores

    def rank_device(
        self,
        subject_ids: tp.Sequence[int],
        k: tp.Optional[int] = None,
        filter_pairs_csr: tp.Union[None, sparse.csr_matrix, DeviceCSR] = None,
        sorted_object_whitelist: tp.Optional[np.ndarray] = None,
        settle: bool = True,
    ) -> tp.Tuple[torch.Tensor, torch.Tensor, torch.Tensor, np.ndarray]:
        """Device-resident result: (ids [S,k] int64, scores [S,k] f32, counts [S] int32, subject_ids).

        `filter_pairs_csr` may be a `DeviceCSR` prepared once (no per-call upload).  The two-stage path reads its per-user proof
        flags on the host before it returns (the one synchronisation of a call); `settle=False` skips that read — calls then queue
        back to back on the stream, and `settle()` reads the flags of all of them at once before anybody consumes the results.
        """
        subject_ids = np.asarray(subject_ids)
        if filter_pairs_csr is not None and filter_pairs_csr.shape[0] != len(subject_ids):
            explanation = "Number of rows in `filter_pairs_csr` must be equal to `len(sublect_ids)`"
            raise ValueError(explanation)

        dev = self.device
        n_objects = self.objects_factors.shape[0]
        whitelist_t = None
        id_offset = 0
        if sorted_object_whitelist is not None:
            wl = np.asarray(sorted_object_whitelist, dtype=np.int64)
            n_cand = len(wl)
            if n_cand > 0 and int(wl[-1]) - int(wl[0]) == n_cand - 1 and (n_cand < 3 or bool(np.all(np.diff(wl) == 1))):
                id_offset = int(wl[0])  # contiguous range (the recommend() default): no indirection needed
            elif n_cand > 0:
                whitelist_t = torch.from_numpy(wl).to(dev)
        else:
            n_cand = n_objects
        if k is None:
            k = n_cand
        kk = min(int(k), n_cand)
        n_subj = len(subject_ids)

        ids_t = torch.empty((n_subj, max(kk, 0)), dtype=torch.int64, device=dev)
        scores_t = torch.empty((n_subj, max(kk, 0)), dtype=torch.float32, device=dev)
        counts_t = torch.zeros((n_subj,), dtype=torch.int32, device=dev)
        if n_subj == 0 or kk <= 0:
            return ids_t, scores_t, counts_t, subject_ids

        rows_t = None
        if not (n_subj == self.subjects_factors.shape[0] and np.array_equal(subject_ids, np.arange(n_subj))):
            rows_t = torch.from_numpy(subject_ids.astype(np.int64)).to(dev)

        indptr_t = indices_t = hash_t = None
        if filter_pairs_csr is not None:
            dcsr = filter_pairs_csr if isinstance(filter_pairs_csr, DeviceCSR) else DeviceCSR.from_scipy(
                filter_pairs_csr, dev)
            indptr_t, indices_t = dcsr.indptr, dcsr.indices
            # many users against a small catalog: the per-block selection slow path dominates, give it O(1) lookups
            if n_subj * 64 >= n_cand:
                hash_t = dcsr.hash_tables()

        upp = 0 if self.batch_size is None else int(self.batch_size)
        if self._two_stage_applies(kk, n_cand, n_subj):
            self._rank_two_stage(ids_t, scores_t, counts_t, rows_t, n_subj, whitelist_t, n_cand, id_offset, kk, indptr_t, indices_t,
                                 hash_t, upp, settle=settle)
        else:
            self._rank_exact(ids_t, scores_t, counts_t, rows_t, 0, n_subj, whitelist_t, n_cand, id_offset, kk, indptr_t, indices_t,
                             hash_t, upp)
        return ids_t, scores_t, counts_t, subject_ids
