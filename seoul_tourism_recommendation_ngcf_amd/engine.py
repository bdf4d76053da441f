"""Host side of the propagation engine: thin, typed wrappers over the C ABI (include/ngcf_hip.h).

Everything here hands `tensor.data_ptr()` + the current HIP stream to libngcf_hip.so.  torch is
used for device memory and streams only; no tensor arithmetic of the hot path happens in torch.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import torch

from . import _lib

LEAKY_SLOPE = 0.2            # NGCF.py:140


def _require_device(t: torch.Tensor, what: str):
    if not t.is_cuda:
        raise RuntimeError(
            f"{what} is on '{t.device}': the NGCF propagation engine runs on a ROCm device only "
            "(hand-written HIP kernels, no CPU/PyTorch fallback). Move the module and inputs to 'cuda'.")


def _f32c(t: torch.Tensor, what: str) -> torch.Tensor:
    _require_device(t, what)
    if t.dtype != torch.float32:
        raise RuntimeError(f"{what}: expected float32, got {t.dtype}")
    return t


def _ptr(t: Optional[torch.Tensor]):
    return C.c_void_p(0 if t is None else t.data_ptr())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


class _NullCtx:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


_NULL = _NullCtx()


def _on(device):
    """`torch.cuda.device(device)`, or nothing at all when that device is already current (the context manager costs ~8 us per
    use, a dozen times per forward on a launch-bound graph)."""
    dev = torch.device(device)
    if dev.index is None or dev.index == torch.cuda.current_device():
        return _NULL
    return torch.cuda.device(dev)


def _stream():
    """torch's current stream on the current device as a hipStream_t.  (The raw getter where this torch has it: the Stream
    object of `torch.cuda.current_stream()` costs ~8 us to build, seven times per forward on a launch-bound graph.)"""
    if _raw_stream is not None:
        return C.c_void_p(_raw_stream(torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _row_major_ld(t: torch.Tensor, what: str) -> int:
    """Leading dimension of a 2-D row-major (possibly column-sliced) fp32 tensor."""
    if t.dim() != 2 or (t.shape[1] > 1 and t.stride(1) != 1):
        raise RuntimeError(f"{what}: expected a row-major 2-D tensor, got shape {tuple(t.shape)} strides {t.stride()}")
    return int(t.stride(0)) if t.shape[0] > 1 else max(int(t.stride(0)), int(t.shape[1]))


class LaplacianCSR:
    """Device CSR of one Laplacian slice (or a row slab of it), built from the COO of `lap_list[k]`.

    Replaces `self.lap_list[year_idx].to(self.device)` + the COO SpMM set-up of NGCF.py:118,130.
    """

    def __init__(self, handle: int, keep_alive=()):
        self._h = C.c_void_p(handle)
        self._keep = keep_alive
        lib = _lib.load()
        self.n_rows = int(lib.ngcf_csr_n_rows(self._h))
        self.n_cols = int(lib.ngcf_csr_n_cols(self._h))
        self.nnz = int(lib.ngcf_csr_nnz(self._h))

    @property
    def max_row_len(self) -> int:
        """Stored entries of the longest row."""
        return int(_lib.load().ngcf_csr_max_row_len(self._h))

    # -- constructors ---------------------------------------------------------------------
    @classmethod
    def from_coo(cls, rows: torch.Tensor, cols: torch.Tensor, vals: torch.Tensor, n_rows: int, n_cols: int):
        lib = _lib.load()
        for t, nm in ((rows, "rows"), (cols, "cols"), (vals, "vals")):
            _require_device(t, "Laplacian " + nm)
        rows = rows.contiguous().to(torch.int64)
        cols = cols.contiguous().to(torch.int64)
        vals = vals.contiguous().to(torch.float32)
        if not (rows.numel() == cols.numel() == vals.numel()):
            raise RuntimeError("Laplacian COO arrays differ in length")
        out = C.c_void_p()
        with _on(vals.device):
            _lib.check(lib.ngcf_csr_from_coo(_ptr(rows), _ptr(cols), _ptr(vals), rows.numel(), n_rows, n_cols,
                                             C.byref(out), _stream()))
        return cls(out.value)

    @classmethod
    def from_sparse_coo(cls, L: torch.Tensor, device, row_range=None):
        """From a torch sparse COO tensor (the element type of `lap_list`, matrix.py:79-83)."""
        if not L.is_sparse:
            raise RuntimeError("lap_list entries must be torch sparse COO tensors (matrix.py:79-83)")
        idx = L._indices().to(device)
        val = L._values().to(device=device, dtype=torch.float32)
        n_rows, n_cols = int(L.shape[0]), int(L.shape[1])
        rows, cols = idx[0], idx[1]
        if row_range is not None:
            lo, hi = row_range
            sel = (rows >= lo) & (rows < hi)
            rows, cols, val = rows[sel] - lo, cols[sel], val[sel]
            n_rows = hi - lo
        return cls.from_coo(rows, cols, val, n_rows, n_cols)

    @classmethod
    def from_csr_arrays(cls, rowptr: torch.Tensor, colidx: torch.Tensor, vals: torch.Tensor, n_cols: int):
        lib = _lib.load()
        for t, nm in ((rowptr, "rowptr"), (colidx, "colidx"), (vals, "vals")):
            _require_device(t, "CSR " + nm)
        assert rowptr.dtype == torch.int64 and colidx.dtype == torch.int32 and vals.dtype == torch.float32
        rowptr, colidx, vals = rowptr.contiguous(), colidx.contiguous(), vals.contiguous()
        out = C.c_void_p()
        with _on(vals.device):
            _lib.check(lib.ngcf_csr_from_arrays(_ptr(rowptr), _ptr(colidx), _ptr(vals), rowptr.numel() - 1, n_cols,
                                                colidx.numel(), C.byref(out), _stream()))
        return cls(out.value, keep_alive=(rowptr, colidx, vals))

    def filtered(self, keep: torch.Tensor, entry_map: Optional[torch.Tensor] = None, nnz_kept: int = -1,
                 reuse: Optional["LaplacianCSR"] = None) -> "LaplacianCSR":
        """Thinned copy on the device (ngcf_csr_filter): the entries e with keep[entry_map[e]] (entry_map None: keep[e]) in this
        matrix's order - the reference's `sparse_dropout` (NGCF.py:93-100) without a COO rebuild, a host round trip or, when
        `reuse` (the object a previous call returned for the same source shape) is given, an allocation.  `keep`: device uint8 /
        bool; `entry_map`: device int32.  The copy borrows this object's segment lists and keeps it alive."""
        lib = _lib.load()
        _require_device(keep, "keep flags")
        if keep.dtype not in (torch.uint8, torch.bool) or not keep.is_contiguous():
            raise RuntimeError("filtered: keep must be a contiguous uint8 / bool tensor")
        if entry_map is not None and (entry_map.dtype != torch.int32 or not entry_map.is_contiguous() or entry_map.numel() != self.nnz):
            raise RuntimeError("filtered: entry_map must be a contiguous int32 tensor with one element per stored entry")
        if entry_map is None and keep.numel() != self.nnz:
            raise RuntimeError(f"filtered: {keep.numel()} keep flags for {self.nnz} stored entries")
        out = C.c_void_p(reuse._h.value if reuse is not None and reuse._h.value else None)
        with _on(keep.device):
            rc = lib.ngcf_csr_filter(self._h, _ptr(keep), _ptr(entry_map), int(nnz_kept), C.byref(out), _stream())
        if reuse is not None:                              # the handle moved into the returned object (or was replaced by the library)
            reuse._h = C.c_void_p(0 if rc == _lib.OK else (out.value or 0))
        _lib.check(rc)
        res = LaplacianCSR(out.value, keep_alive=(self,))
        res.src_nnz = self.nnz
        return res

    @property
    def filter_pos(self) -> int:
        """Device address of the int32[source nnz + 1] scan `pos` of a filtered copy (position of every kept entry)."""
        return int(_lib.load().ngcf_csr_filter_pos(self._h) or 0)

    # -- misc -----------------------------------------------------------------------------
    def plan(self, seg_len: int):
        _lib.check(_lib.load().ngcf_csr_plan(self._h, int(seg_len), _stream()))

    def set_mode(self, mode: int):
        """0 row-wise kernels (d-sliced where it pays), 1 row-wise without slicing, 2 L2-swept kernel wherever the shape
        allows (tests), 3 L2-swept kernel on the row groups where it is expected to pay (long-lived matrices)."""
        _lib.check(_lib.load().ngcf_csr_set_mode(self._h, int(mode), _stream()))

    @property
    def n_segments(self) -> int:
        return int(_lib.load().ngcf_csr_n_segments(self._h))

    @property
    def swept_rows(self) -> int:
        """Rows covered by L2-swept parts (0: all products of this CSR use the row-wise kernels)."""
        return int(_lib.load().ngcf_csr_swept_rows(self._h))

    def layer_workspace_bytes(self, d_in: int, d_out: int) -> int:
        n = int(_lib.load().ngcf_layer_workspace_bytes(self._h, d_in, d_out))
        if n < 0:
            raise RuntimeError(f"unsupported layer widths d_in={d_in} d_out={d_out} (1..512)")
        return n

    def spmm_workspace_bytes(self, d: int) -> int:
        return int(_lib.load().ngcf_spmm_workspace_bytes(self._h, d))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            _lib.load().ngcf_csr_free(self._h)
            self._h = C.c_void_p(0)

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001  (interpreter shutdown)
            pass


class Workspace:
    """A grow-only byte buffer on one device, handed to the kernels as scratch."""

    def __init__(self):
        self.buf: Optional[torch.Tensor] = None

    def get(self, nbytes: int, device) -> torch.Tensor:
        if self.buf is None or self.buf.numel() < nbytes or self.buf.device != torch.device(device):
            self.buf = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
        return self.buf


def spmm(csr: LaplacianCSR, E: torch.Tensor, out: Optional[torch.Tensor] = None, ws: Optional[Workspace] = None,
         edge_drop=None):
    """LE = L.E (NGCF.py:130) through ngcf_spmm_csr_f32.

    `edge_drop = (seeds, p, transposed)` applies device-side node dropout (ngcf_spmm_csr_dropout_f32): `seeds` the
    cumulative list of 64-bit layer seeds, `p` the drop probability, `transposed` true when `csr` holds L^T (the mask is
    keyed by an entry's row and column in L, so L^T loses the entries L lost)."""
    lib = _lib.load()
    _f32c(E, "E")
    d = int(E.shape[1])
    if E.shape[0] != csr.n_cols:
        raise RuntimeError(f"mat1 and mat2 shapes cannot be multiplied ({csr.n_rows}x{csr.n_cols} and {tuple(E.shape)})")
    d_view = d
    if out is None:     # rows padded to a multiple of 32 floats (128-byte aligned rows: the float4 / swept kernels apply at any d)
        out = torch.empty((csr.n_rows, (d + 31) // 32 * 32), dtype=torch.float32, device=E.device)[:, :d]
        # the rule of ngcf_layer_fused_f32 (csrc/dense.hip), so that both forward paths produce the same bits: a width that is
        # not a multiple of 4 on a small (launch-bound) matrix is multiplied up to the next multiple of 4 when the gathered rows
        # are 16-byte aligned and padded - the extra columns land in the padding of `out`
        d = int(lib.ngcf_spmm_product_width(csr._h, _ptr(E), _row_major_ld(E, "E"), d))
    ws = ws or Workspace()
    nb = csr.spmm_workspace_bytes(d)
    w = ws.get(nb, E.device)
    with _on(E.device):
        if edge_drop is None:
            _lib.check(lib.ngcf_spmm_csr_f32(csr._h, _ptr(E), _row_major_ld(E, "E"), d, _ptr(out),
                                             _row_major_ld(out, "out"), _ptr(w), w.numel(), _stream()))
        else:
            seeds, p, transposed = edge_drop
            arr = (C.c_uint64 * max(len(seeds), 1))(*[int(x) & (2 ** 64 - 1) for x in seeds])
            _lib.check(lib.ngcf_spmm_csr_dropout_f32(csr._h, _ptr(E), _row_major_ld(E, "E"), d, _ptr(out),
                                                     _row_major_ld(out, "out"), float(p), arr, len(seeds), 1 if transposed else 0,
                                                     _ptr(w), w.numel(), _stream()))
    assert out.shape[1] == d_view
    return out


def spmm_t_rows(csr_t: LaplacianCSR, slot: torch.Tensor, X: torch.Tensor, init: Optional[torch.Tensor], out: torch.Tensor,
                ws: Workspace, edge_drop=None):
    """out = init + L^T . X for a row-sparse X given compacted (ngcf_spmm_t_rows_f32): `slot` int32[N] maps a matrix row to its
    row of X / init or -1; every row of `out` is written, every sum runs in a fixed order (no atomics).  `edge_drop = (seeds, p)`:
    device-side node dropout of the forward product."""
    lib = _lib.load()
    _f32c(X, "X"), _f32c(out, "out")
    if slot.dtype != torch.int32 or slot.numel() != csr_t.n_cols or out.shape[0] != csr_t.n_rows or X.shape[1] != out.shape[1]:
        raise RuntimeError("spmm_t_rows: shape mismatch")
    if init is not None and tuple(init.shape) != tuple(X.shape):
        raise RuntimeError("spmm_t_rows: init must have the shape of X")
    seeds, p = edge_drop if edge_drop is not None else ((), 0.0)
    arr = (C.c_uint64 * max(len(seeds), 1))(*[int(x) & (2 ** 64 - 1) for x in seeds])
    d = int(X.shape[1])
    w = ws.get(csr_t.spmm_workspace_bytes(min(d, 512)), X.device)
    with _on(X.device):
        _lib.check(lib.ngcf_spmm_t_rows_f32(csr_t._h, _ptr(slot), _ptr(X), _row_major_ld(X, "X"), d, _ptr(init),
                                            0 if init is None else _row_major_ld(init, "init"), _ptr(out), _row_major_ld(out, "out"),
                                            float(p), arr, len(seeds), _ptr(w), w.numel(), _stream()))


def layer_fused(csr: LaplacianCSR, E_gather: torch.Tensor, E_self: torch.Tensor, W1, b1, W2, b2,
                carry: Optional[torch.Tensor], norm: torch.Tensor, ws: Workspace,
                drop_p: float = 0.0, drop_seed: int = 0, drop_mask: Optional[torch.Tensor] = None):
    """One propagation layer (NGCF.py:130-146) through ngcf_layer_fused_f32.  `drop_mask` [n_rows, d_out]: the noise
    tensor of nn.Dropout (0 or 1/(1-p)) drawn by the caller; None with drop_p > 0: the in-kernel hash stream."""
    lib = _lib.load()
    d_in, d_out = int(W1.shape[1]), int(W1.shape[0])
    for t, nm in ((E_gather, "E_gather"), (E_self, "E_self"), (W1, "W1"), (b1, "b1"), (W2, "W2"), (b2, "b2"), (norm, "norm")):
        _f32c(t, nm)
    if E_gather.shape[1] != d_in or E_self.shape[1] != d_in:
        raise RuntimeError(f"mat1 and mat2 shapes cannot be multiplied ({tuple(E_self.shape)} and {d_in}x{d_out})")
    if E_gather.shape[0] != csr.n_cols or E_self.shape[0] != csr.n_rows or norm.shape[0] != csr.n_rows:
        raise RuntimeError("layer_fused: row counts do not match the Laplacian")
    W1, W2, b1, b2 = W1.contiguous(), W2.contiguous(), b1.contiguous(), b2.contiguous()
    nb = csr.layer_workspace_bytes(d_in, d_out)
    w = ws.get(nb, norm.device)
    with _on(norm.device):
        _lib.check(lib.ngcf_layer_fused_f32(
            csr._h, _ptr(E_gather), _row_major_ld(E_gather, "E_gather"), _ptr(E_self), _row_major_ld(E_self, "E_self"),
            d_in, _ptr(W1), _ptr(b1), _ptr(W2), _ptr(b2), d_out, LEAKY_SLOPE, float(drop_p), int(drop_seed),
            _ptr(drop_mask), 0 if drop_mask is None else _row_major_ld(drop_mask, "drop_mask"),
            _ptr(carry), 0 if carry is None else _row_major_ld(carry, "carry"),
            _ptr(norm), _row_major_ld(norm, "norm"), _ptr(w), w.numel(), _stream()))


def layer_dense(LE: torch.Tensor, E_self: torch.Tensor, W1, b1, W2, b2, carry, norm, ws: Workspace,
                drop_p: float = 0.0, drop_seed: int = 0, drop_mask: Optional[torch.Tensor] = None):
    """Dense half of a layer (NGCF.py:131-146) on an existing LE, through ngcf_layer_dense_f32."""
    lib = _lib.load()
    d_in, d_out = int(W1.shape[1]), int(W1.shape[0])
    W1, W2, b1, b2 = W1.contiguous(), W2.contiguous(), b1.contiguous(), b2.contiguous()
    nb = int(lib.ngcf_dense_workspace_bytes(d_in, d_out))
    if nb < 0:
        raise RuntimeError(f"unsupported layer widths d_in={d_in} d_out={d_out}")
    w = ws.get(nb, norm.device)
    with _on(norm.device):
        _lib.check(lib.ngcf_layer_dense_f32(
            _ptr(LE), _row_major_ld(LE, "LE"), _ptr(E_self), _row_major_ld(E_self, "E_self"), LE.shape[0], d_in,
            _ptr(W1), _ptr(b1), _ptr(W2), _ptr(b2), d_out, LEAKY_SLOPE, float(drop_p), int(drop_seed),
            _ptr(drop_mask), 0 if drop_mask is None else _row_major_ld(drop_mask, "drop_mask"),
            _ptr(carry), 0 if carry is None else _row_major_ld(carry, "carry"),
            _ptr(norm), _row_major_ld(norm, "norm"), _ptr(w), w.numel(), _stream()))


def copy_rows(src: torch.Tensor, dst: torch.Tensor, dst2: Optional[torch.Tensor] = None):
    """dst[:, :] = src (strided row copy, ngcf_copy_rows_f32); with `dst2` also dst2[:, :] = src in the same pass."""
    lib = _lib.load()
    _f32c(src, "src"), _f32c(dst, "dst")
    if src.shape != dst.shape or (dst2 is not None and dst2.shape != src.shape):
        raise RuntimeError(f"copy_rows: shape mismatch {tuple(src.shape)} vs {tuple(dst.shape)}")
    if src.shape[0] == 0:
        return
    with _on(dst.device):
        if dst2 is None:
            _lib.check(lib.ngcf_copy_rows_f32(_ptr(src), _row_major_ld(src, "src"), _ptr(dst), _row_major_ld(dst, "dst"),
                                              src.shape[0], src.shape[1], _stream()))
        else:
            _lib.check(lib.ngcf_copy_rows2_f32(_ptr(src), _row_major_ld(src, "src"), _ptr(dst), _row_major_ld(dst, "dst"),
                                               _ptr(_f32c(dst2, "dst2")), _row_major_ld(dst2, "dst2"), src.shape[0],
                                               src.shape[1], _stream()))


def copy_rows_indexed(src: torch.Tensor, dst: torch.Tensor, idx: torch.Tensor):
    """dst[idx[b], :] = src[idx[b], :] (ngcf_copy_rows_indexed_f32): the rows a batch touched, ids out of range skipped."""
    lib = _lib.load()
    _f32c(src, "src"), _f32c(dst, "dst")
    if src.shape != dst.shape or idx.dtype != torch.int64 or not idx.is_contiguous() or idx.device != dst.device:
        raise RuntimeError("copy_rows_indexed: src and dst must have one shape, idx must be a contiguous int64 tensor on their device")
    if idx.numel() == 0 or src.shape[0] == 0:
        return
    with _on(dst.device):
        _lib.check(lib.ngcf_copy_rows_indexed_f32(_ptr(src), _row_major_ld(src, "src"), _ptr(dst), _row_major_ld(dst, "dst"), _ptr(idx),
                                                  idx.numel(), src.shape[0], src.shape[1], _stream()))


def gather_rows(table: torch.Tensor, idx: torch.Tensor, status: torch.Tensor, row_off: int = 0,
                n_idx_rows: Optional[int] = None) -> torch.Tensor:
    """out[b] = table[row_off + idx[b]] (NGCF.py:151-155), bit-exact copies; fresh output tensor."""
    lib = _lib.load()
    _f32c(table, "table")
    idx = idx.to(device=table.device, dtype=torch.int64).contiguous()
    d = int(table.shape[1])
    B = int(idx.numel())
    out = torch.empty((B, d), dtype=torch.float32, device=table.device)
    if n_idx_rows is None:
        n_idx_rows = int(table.shape[0]) - row_off
    with _on(table.device):
        _lib.check(lib.ngcf_gather_rows_f32(_ptr(table), _row_major_ld(table, "table"), d, _ptr(idx), B, row_off,
                                            n_idx_rows, _ptr(out), d, _ptr(status), _stream()))
    return out


def gather_rows3(table: torch.Tensor, sets, status: torch.Tensor):
    """The (users, positive items, negative items) gathers of NGCF.py:151-155 in one launch.  `sets` = three
    `(idx or None, row_off, n_idx_rows)`; returns three fresh tensors (`None` where idx is None).  Bit-exact copies."""
    lib = _lib.load()
    _f32c(table, "table")
    d = int(table.shape[1])
    args, outs = [], []
    for idx, row_off, n_rows in sets:
        if idx is None:
            args += [None, 0, 0, 0, None]
            outs.append(None)
            continue
        idx = idx.to(device=table.device, dtype=torch.int64).contiguous()
        out = torch.empty((int(idx.numel()), d), dtype=torch.float32, device=table.device)
        args += [_ptr(idx), int(idx.numel()), int(row_off), int(n_rows), _ptr(out)]
        outs.append(out)
    with _on(table.device):
        _lib.check(lib.ngcf_gather_rows3_f32(_ptr(table), _row_major_ld(table, "table"), d, *args, d, _ptr(status), _stream()))
    return outs


def feature_inject(user_w: torch.Tensor, tables: Sequence[torch.Tensor], idx: Sequence[torch.Tensor],
                   u_id: torch.Tensor, emb_ratio: float, scratch: torch.Tensor, status: torch.Tensor):
    """user_w[u_id] = user_w[u_id]*(1-r) + cat(feature rows)*r in place (NGCF.py:103-115)."""
    lib = _lib.load()
    _f32c(user_w, "user_embedding.weight")
    if not user_w.is_contiguous():
        raise RuntimeError("user_embedding.weight must be contiguous")
    dev = user_w.device
    tabs = [_f32c(t, "feature table").contiguous() for t in tables]
    ids = [i.to(device=dev, dtype=torch.int64).contiguous() for i in idx]
    u_id = u_id.to(device=dev, dtype=torch.int64).contiguous()
    B = int(u_id.numel())
    for i in ids:
        if int(i.numel()) != B:
            raise RuntimeError("shape mismatch: feature index vectors and u_id differ in length")
    fw = int(tabs[0].shape[1])
    t_arr = (C.c_void_p * 5)(*[t.data_ptr() for t in tabs])
    i_arr = (C.c_void_p * 5)(*[i.data_ptr() for i in ids])
    c_arr = (C.c_int64 * 5)(*[int(t.shape[0]) for t in tabs])
    with _on(dev):
        _lib.check(lib.ngcf_feature_inject_f32(_ptr(user_w), user_w.shape[1], user_w.shape[0], user_w.shape[1],
                                               t_arr, i_arr, c_arr, fw, _ptr(u_id), B, float(emb_ratio),
                                               _ptr(scratch), _ptr(status), _stream()))
    return ids  # keep the converted index tensors alive until the stream has consumed them


def bpr_loss(u: torch.Tensor, p: torch.Tensor, n: torch.Tensor, weight_decay: float, batch_size: float,
             ws: Workspace) -> torch.Tensor:
    """Fused BPR (bprloss.py:15-22) through ngcf_bpr_fused_f32 -> 0-dim device tensor."""
    lib = _lib.load()
    for t, nm in ((u, "u"), (p, "pos"), (n, "neg")):
        _f32c(t, nm)
        if t.dim() != 2:
            raise RuntimeError(f"BPR: {nm} must be 2-D, got shape {tuple(t.shape)}")
    D = int(u.shape[1])
    if p.shape[1] != D or n.shape[1] != D:
        raise RuntimeError(f"The size of tensor a ({D}) must match the size of tensor b ({p.shape[1]}/{n.shape[1]}) "
                           "at non-singleton dimension 1")
    u, p, n = u.contiguous(), p.contiguous(), n.contiguous()
    R = max(u.shape[0], p.shape[0], n.shape[0])
    nb = int(lib.ngcf_bpr_workspace_bytes(R))
    w = ws.get(nb, u.device)
    loss = torch.empty((), dtype=torch.float32, device=u.device)
    with _on(u.device):
        _lib.check(lib.ngcf_bpr_fused_f32(_ptr(u), u.shape[0], _ptr(p), p.shape[0], _ptr(n), n.shape[0], D,
                                          float(weight_decay), float(batch_size), _ptr(loss), _ptr(w), w.numel(),
                                          _stream()))
    return loss


def topk_rows(scores: torch.Tensor, k: int):
    """`torch.topk(scores, k)` for a 2-D fp32 score matrix (values, int64 indices), through ngcf_topk_rows_f32."""
    lib = _lib.load()
    _f32c(scores, "scores")
    if scores.dim() != 2:
        raise RuntimeError("topk_rows: expected a 2-D score matrix")
    n_rows, n_cols = int(scores.shape[0]), int(scores.shape[1])
    scores = scores if scores.stride(1) == 1 else scores.contiguous()
    vals = torch.empty((n_rows, k), dtype=torch.float32, device=scores.device)
    idx = torch.empty((n_rows, k), dtype=torch.int64, device=scores.device)
    with _on(scores.device):
        _lib.check(lib.ngcf_topk_rows_f32(_ptr(scores), _row_major_ld(scores, "scores"), n_rows, n_cols, int(k), _ptr(vals),
                                          _ptr(idx), _stream()))
    return vals, idx


def recommend_topk(u_emb: torch.Tensor, item_emb: torch.Tensor, k: int, return_scores: bool = False):
    """Scores of every item for every user row and their top-k (experiment.py:93,104-111; demo.py:233-235) in ONE launch of a
    hand-written kernel (ngcf_recommend_topk_f32): score tiles through LDS, then radix select + bitonic sort per row.
    Returns (values [B, k], int64 indices [B, k]); with `return_scores` also the [B, n_items] score matrix (what
    `torch.mm(u_emb, item_emb.T)` is in the reference)."""
    lib = _lib.load()
    _f32c(u_emb, "u_emb"), _f32c(item_emb, "item_emb")
    if u_emb.dim() != 2 or item_emb.dim() != 2 or u_emb.shape[1] != item_emb.shape[1]:
        raise RuntimeError(f"mat1 and mat2 shapes cannot be multiplied ({tuple(u_emb.shape)} and {tuple(item_emb.t().shape)})")
    B, D, n_items = int(u_emb.shape[0]), int(u_emb.shape[1]), int(item_emb.shape[0])
    u_emb = u_emb if u_emb.stride(1) == 1 else u_emb.contiguous()
    item_emb = item_emb if item_emb.stride(1) == 1 else item_emb.contiguous()
    scores = torch.empty((B, n_items), dtype=torch.float32, device=u_emb.device)
    vals = torch.empty((B, k), dtype=torch.float32, device=u_emb.device)
    idx = torch.empty((B, k), dtype=torch.int64, device=u_emb.device)
    with _on(u_emb.device):
        _lib.check(lib.ngcf_recommend_topk_f32(_ptr(u_emb), _row_major_ld(u_emb, "u_emb"), B, _ptr(item_emb),
                                               _row_major_ld(item_emb, "item_emb"), n_items, D, int(k), _ptr(scores), n_items,
                                               _ptr(vals), _ptr(idx), _stream()))
    return (vals, idx, scores) if return_scores else (vals, idx)


def shard_plan(rowptr_host: torch.Tensor, row_begin: int, row_end: int, world: int):
    """nnz-balanced contiguous cut of rows [row_begin,row_end) into `world` ranges (host helper)."""
    lib = _lib.load()
    rp = rowptr_host.to(device="cpu", dtype=torch.int64).contiguous()
    bounds = (C.c_int64 * (world + 1))()
    _lib.check(lib.ngcf_shard_plan(C.cast(rp.data_ptr(), C.POINTER(C.c_int64)), row_begin, row_end, world, bounds))
    return [int(b) for b in bounds]
