"""Device plumbing above the C ABI: a context bound to a torch CUDA(HIP) device/stream, device
tensors (torch is used ONLY as the allocator / stream / collective provider -- all arithmetic is
libmpsk's hand-written HIP) and thin typed wrappers over the mpsk_* entry points.

Layout: every tensor is a flat fp64 torch tensor holding the column-major (TensorKit) data;
`DTensor.shape` carries the logical index order, e.g. (Dl, d, Dr).  Environments are stored as
(W, Dbra, Dket) = W column-major slabs (include/mpsk.h).
"""
from __future__ import annotations

import ctypes as C
import numpy as np

from . import _lib
from ._lib import MpskError, check


def _torch():
    import torch
    return torch


class DTensor:
    """Flat device buffer + logical column-major shape."""
    __slots__ = ("buf", "shape")

    def __init__(self, buf, shape):
        self.buf = buf
        self.shape = tuple(int(s) for s in shape)

    @property
    def ptr(self):
        return self.buf.data_ptr()

    @property
    def size(self):
        n = 1
        for s in self.shape:
            n *= s
        return n

    def reshape(self, *shape):
        if len(shape) == 1 and isinstance(shape[0], (tuple, list)):
            shape = tuple(shape[0])
        t = DTensor(self.buf, shape)
        assert t.size == self.size, (self.shape, shape)
        return t


class Backend:
    """One mpsk_ctx on one GPU.  Raises MpskError when the HIP library or a GPU is missing."""

    def __init__(self, device=0, stream=None):
        """stream: a torch.cuda.Stream to bind (default: the calling thread's current torch stream).  Two Backends on
        two streams may be driven from two host threads at the same time (include/mpsk.h, thread-safety note)."""
        torch = _torch()
        if not torch.cuda.is_available():
            raise MpskError("no HIP device visible: the mpskit.jl_amd product path has no CPU fallback")
        self.lib = _lib.load()
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        h = C.c_void_p()
        check(self.lib.mpsk_ctx_create(device, C.byref(h)), "mpsk_ctx_create")
        self.ctx = h
        self.bind_stream(stream)

    def bind_stream(self, stream=None):
        torch = _torch()
        self.torch_stream = torch.cuda.current_stream(self.device) if stream is None else stream
        check(self.lib.mpsk_ctx_set_stream(self.ctx, C.c_void_p(self.torch_stream.cuda_stream)), "mpsk_ctx_set_stream")

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.mpsk_ctx_destroy(self.ctx)
            self.ctx = None

    def synchronize(self):
        check(self.lib.mpsk_ctx_synchronize(self.ctx), "mpsk_ctx_synchronize")

    def set_qr_mode(self, mode):
        check(self.lib.mpsk_ctx_set_qr_mode(self.ctx, int(mode)), "mpsk_ctx_set_qr_mode")

    def qr_retries(self) -> int:
        """CholeskyQR3 factorizations repeated with the published shift after a breakdown of the rounding-level one."""
        r = C.c_long(0)
        check(self.lib.mpsk_ctx_qr_retries(self.ctx, C.byref(r)), "mpsk_ctx_qr_retries")
        return r.value

    def qr_stats(self):
        a, b, f, r = C.c_long(), C.c_long(), C.c_long(), C.c_long()
        check(self.lib.mpsk_ctx_qr_stats(self.ctx, C.byref(a), C.byref(b), C.byref(f), C.byref(r)), "mpsk_ctx_qr_stats")
        return {"cholqr3": a.value, "householder": b.value, "fallback": f.value, "robust": r.value}

    def set_svd_mode(self, precondition=True):
        """0 / False: Jacobi on theta itself; 1 / True: QR-preconditioned; 2: QR + QR of R^T; 3: 2 + truncation-aware
        mpsk_tsplit (subspace iteration, checked; include/mpsk.h)."""
        check(self.lib.mpsk_ctx_set_svd_mode(self.ctx, int(precondition)), "mpsk_ctx_set_svd_mode")

    def svd_sweeps(self):
        n = C.c_int()
        check(self.lib.mpsk_ctx_svd_stats(self.ctx, C.byref(n)), "mpsk_ctx_svd_stats")
        return n.value

    def split_stats(self):
        """last mpsk_tsplit: {"path": 0 full iteration / 1 subspace stage accepted / 2 stage gave up, "iterations", "residual"}"""
        p, it, r = C.c_int(), C.c_int(), C.c_double()
        check(self.lib.mpsk_ctx_split_stats(self.ctx, C.byref(p), C.byref(it), C.byref(r)), "mpsk_ctx_split_stats")
        return {"path": p.value, "iterations": it.value, "residual": r.value}

    def prof_enable(self, on=True):
        check(self.lib.mpsk_prof_enable(self.ctx, int(on)), "mpsk_prof_enable")

    def prof_summary(self):
        import json
        buf = C.create_string_buffer(1 << 16)
        check(self.lib.mpsk_prof_summary(self.ctx, buf, len(buf)), "mpsk_prof_summary")
        return json.loads(buf.value.decode())

    # ---- memory ---------------------------------------------------------------------------
    def empty(self, *shape):
        torch = _torch()
        if len(shape) == 1 and isinstance(shape[0], (tuple, list)):
            shape = tuple(shape[0])
        n = 1
        for s in shape:
            n *= int(s)
        return DTensor(torch.empty(max(n, 1), dtype=torch.float64, device=self.device), shape)

    def zeros(self, *shape):
        t = self.empty(*shape)
        t.buf.zero_()
        return t

    def upload(self, a, shape=None):
        """host ndarray (logical index order) -> device column-major."""
        torch = _torch()
        a = np.asarray(a)
        if np.iscomplexobj(a):
            if np.abs(a.imag).max(initial=0.0) > 0:
                raise MpskError("complex128 tensors are not supported by the fp64 device path (MPSK_C128 reserved)")
            a = a.real
        a = np.ascontiguousarray(a, dtype=np.float64)
        flat = np.ravel(a, order="F")
        buf = torch.from_numpy(np.ascontiguousarray(flat)).to(self.device)
        if buf.numel() == 0:
            buf = torch.empty(1, dtype=torch.float64, device=self.device)
        return DTensor(buf, a.shape if shape is None else shape)

    def download(self, t: DTensor):
        flat = t.buf[: t.size].cpu().numpy()
        return flat.reshape(t.shape, order="F").copy()

    # ---- complex128 tensors: interleaved (re, im) = a REAL tensor whose first dimension is doubled ("half-embedded"
    # view, include/mpsk.h); the complex operator paths take such DTensors (flag cplx / a complex MPO slice)
    def upload_c(self, a):
        """complex ndarray (logical index order) -> device, interleaved complex128 column-major; shape (2 n0, n1, ...)."""
        torch = _torch()
        a = np.asarray(a, dtype=np.complex128)
        flat = np.ravel(a, order="F").view(np.float64)
        buf = torch.from_numpy(np.ascontiguousarray(flat)).to(self.device)
        return DTensor(buf, (2 * a.shape[0],) + tuple(a.shape[1:]))

    def download_c(self, t: DTensor):
        flat = t.buf[: t.size].cpu().numpy().view(np.complex128)
        return flat.reshape((t.shape[0] // 2,) + tuple(t.shape[1:]), order="F").copy()

    def upload_env_c(self, blocks):
        """complex env = list of [Dbra, chi_i, Dket] arrays -> device (W, 2 Dbra, Dket) interleaved slabs."""
        torch = _torch()
        slabs = [np.asarray(b, dtype=np.complex128)[:, k, :] for b in blocks for k in range(np.asarray(b).shape[1])]
        flat = np.concatenate([np.ravel(s_, order="F") for s_ in slabs]).view(np.float64)
        buf = torch.from_numpy(np.ascontiguousarray(flat)).to(self.device)
        return DTensor(buf, (len(slabs), 2 * slabs[0].shape[0], slabs[0].shape[1]))

    def download_env_c(self, t: DTensor, chis):
        W, Db2, Dk = t.shape
        flat = t.buf[: t.size].cpu().numpy().view(np.complex128)
        Db = Db2 // 2
        slabs = [flat[w * Db * Dk:(w + 1) * Db * Dk].reshape((Db, Dk), order="F") for w in range(W)]
        out, o = [], 0
        for chi in chis:
            out.append(np.stack(slabs[o:o + chi], axis=1))
            o += chi
        return out

    def _set_dtype(self, cplx):
        check(self.lib.mpsk_ctx_set_dtype(self.ctx, 1 if cplx else 0), "mpsk_ctx_set_dtype")

    def upload_env(self, blocks):
        """reference env = list of [Dbra, chi_i, Dket] arrays -> device (W, Dbra, Dket) slabs."""
        slabs = []
        for b in blocks:
            b = np.asarray(b)
            for k in range(b.shape[1]):
                slabs.append(b[:, k, :])
        return self._upload_slabs(slabs)

    def _upload_slabs(self, slabs):
        torch = _torch()
        flat = np.concatenate([np.ravel(np.asarray(s, dtype=np.float64), order="F") for s in slabs])
        buf = torch.from_numpy(flat).to(self.device)
        return DTensor(buf, (len(slabs),) + tuple(slabs[0].shape))

    def download_env(self, t: DTensor, chis):
        W, Db, Dk = t.shape
        flat = t.buf[: t.size].cpu().numpy()
        slabs = [flat[w * Db * Dk:(w + 1) * Db * Dk].reshape((Db, Dk), order="F") for w in range(W)]
        out, o = [], 0
        for chi in chis:
            out.append(np.stack(slabs[o:o + chi], axis=1))
            o += chi
        return out

    def copy(self, t: DTensor):
        return DTensor(t.buf.clone(), t.shape)

    # ---- MPO slices -----------------------------------------------------------------------
    def mposlice(self, odim, d, chil, chir, blocks, cplx=False):
        """blocks: {(i, j): scalar | ndarray [chi_i, d, d, chi_j]} (0-based levels).  cplx: an MPSK_C128 slice (complex
        entries allowed; every operand of a call that takes it is an interleaved complex128 tensor)."""
        return DeviceMPOSlice(self, odim, d, chil, chir, blocks, cplx=cplx)

    # ---- hot-path operators ----------------------------------------------------------------
    def dAC(self, H, GL: DTensor, GR: DTensor, x: DTensor, out: DTensor = None):
        if getattr(H, "cplx", False):          # interleaved complex128 operands: first dimensions are doubled
            Dl2x, d, Dr = x.shape
            Wl, Dlo2, Dl = GL.shape
            assert Dl2x == 2 * Dl and GR.shape == (H.Wr, 2 * Dr, Dr) and Wl == H.Wl and d == H.d, (GL.shape, GR.shape, x.shape)
            y = self.empty(Dlo2, d, Dr) if out is None else out
            check(self.lib.mpsk_dAC(self.ctx, H.handle, Dlo2 // 2, Dl, Dr, GL.ptr, GR.ptr, x.ptr, y.ptr), "mpsk_dAC")
            return y
        Dl, d, Dr = x.shape
        Wl, Dlo, Dl2 = GL.shape
        assert Dl2 == Dl and GR.shape == (H.Wr, Dr, Dr) and Wl == H.Wl and d == H.d, (GL.shape, GR.shape, x.shape)
        y = self.empty(Dlo, d, Dr) if out is None else out
        check(self.lib.mpsk_dAC(self.ctx, H.handle, Dlo, Dl, Dr, GL.ptr, GR.ptr, x.ptr, y.ptr), "mpsk_dAC")
        return y

    def dAC_blocked(self, H, GLrows: DTensor, GR: DTensor, xb: DTensor, nblk, out: DTensor = None):
        """mpsk_dAC_blocked: x in `nblk` row blocks (dist.to_blocked), GLrows = this rank's rows (W, Dlo, Dl);
        returns / fills y[Dlo, d, Dr] (the caller usually points `out` into the blocked destination vector)."""
        Dl, d, Dr = xb.shape
        Wl, Dlo, Dl2 = GLrows.shape
        assert Dl2 == Dl and GR.shape == (H.Wr, Dr, Dr) and Wl == H.Wl and d == H.d and Dl % nblk == 0
        y = self.empty(Dlo, d, Dr) if out is None else out
        check(self.lib.mpsk_dAC_blocked(self.ctx, H.handle, int(nblk), Dlo, Dl, Dr, GLrows.ptr, GR.ptr, xb.ptr, y.ptr),
              "mpsk_dAC_blocked")
        return y

    def hac_create(self, H, GL: DTensor, GR: DTensor):
        """mpsk_hac_create: the prepared effective Hamiltonian of one site (MPO_ddAC).  Returns a PreparedHAC that
        keeps GL / GR alive and releases the device-side object when it is garbage collected."""
        return PreparedHAC(self, H, GL, GR)

    def dC(self, GL: DTensor, GR: DTensor, c: DTensor, out: DTensor = None, cplx=False):
        if cplx:
            Dl2, Dr = c.shape
            W, Dlo2, Dl = GL.shape
            assert Dl2 == 2 * Dl and GR.shape == (W, 2 * Dr, Dr)
            y = self.empty(Dlo2, Dr) if out is None else out
            self._set_dtype(True)
            try:
                check(self.lib.mpsk_dC(self.ctx, W, Dlo2 // 2, Dl, Dr, GL.ptr, GR.ptr, c.ptr, y.ptr), "mpsk_dC")
            finally:
                self._set_dtype(False)
            return y
        Dl, Dr = c.shape
        W, Dlo, _ = GL.shape
        assert GR.shape == (W, Dr, Dr)
        y = self.empty(Dlo, Dr) if out is None else out
        check(self.lib.mpsk_dC(self.ctx, W, Dlo, Dl, Dr, GL.ptr, GR.ptr, c.ptr, y.ptr), "mpsk_dC")
        return y

    def dAC2(self, H1, H2, GL: DTensor, GR: DTensor, x2: DTensor, out: DTensor = None):
        if getattr(H1, "cplx", False):
            Dl2, d1, Dr, d2 = x2.shape
            Wl, Dlo2, Dl = GL.shape
            assert Dl2 == 2 * Dl and GR.shape == (H2.Wr, 2 * Dr, Dr) and Wl == H1.Wl
            y = self.empty(Dlo2, d1, Dr, d2) if out is None else out
            check(self.lib.mpsk_dAC2(self.ctx, H1.handle, H2.handle, Dlo2 // 2, Dl, Dr, GL.ptr, GR.ptr, x2.ptr, y.ptr),
                  "mpsk_dAC2")
            return y
        Dl, d1, Dr, d2 = x2.shape
        Wl, Dlo, _ = GL.shape
        assert GR.shape == (H2.Wr, Dr, Dr) and Wl == H1.Wl
        y = self.empty(Dlo, d1, Dr, d2) if out is None else out
        check(self.lib.mpsk_dAC2(self.ctx, H1.handle, H2.handle, Dlo, Dl, Dr, GL.ptr, GR.ptr, x2.ptr, y.ptr),
              "mpsk_dAC2")
        return y

    def transfer_left(self, H, GLin: DTensor, A: DTensor, Ab: DTensor, out: DTensor = None, cplx=False):
        if cplx or getattr(H, "cplx", False):
            Dl, d, Dr = A.shape[0] // 2, A.shape[1], A.shape[2]
            Dlb, Drb = Ab.shape[0] // 2, Ab.shape[2]
            W = GLin.shape[0]
            Wout = H.Wr if H is not None else W
            y = self.empty(Wout, 2 * Drb, Dr) if out is None else out
            if H is None:
                self._set_dtype(True)
            try:
                check(self.lib.mpsk_transfer_left(self.ctx, H.handle if H is not None else None, W, d, Dl, Dr, Dlb, Drb,
                                                  GLin.ptr, A.ptr, Ab.ptr, y.ptr), "mpsk_transfer_left")
            finally:
                if H is None:
                    self._set_dtype(False)
            return y
        Dl, d, Dr = A.shape
        Dlb, _, Drb = Ab.shape
        W = GLin.shape[0]
        Wout = H.Wr if H is not None else W
        y = self.empty(Wout, Drb, Dr) if out is None else out
        check(self.lib.mpsk_transfer_left(self.ctx, H.handle if H is not None else None, W, d, Dl, Dr, Dlb, Drb,
                                          GLin.ptr, A.ptr, Ab.ptr, y.ptr), "mpsk_transfer_left")
        return y

    def transfer_right(self, H, GRin: DTensor, A: DTensor, Ab: DTensor, out: DTensor = None, cplx=False):
        if cplx or getattr(H, "cplx", False):
            Dl, d, Dr = A.shape[0] // 2, A.shape[1], A.shape[2]
            Dlb, Drb = Ab.shape[0] // 2, Ab.shape[2]
            W = GRin.shape[0]
            Wout = H.Wl if H is not None else W
            y = self.empty(Wout, 2 * Dl, Dlb) if out is None else out
            if H is None:
                self._set_dtype(True)
            try:
                check(self.lib.mpsk_transfer_right(self.ctx, H.handle if H is not None else None, W, d, Dl, Dr, Dlb, Drb,
                                                   A.ptr, Ab.ptr, GRin.ptr, y.ptr), "mpsk_transfer_right")
            finally:
                if H is None:
                    self._set_dtype(False)
            return y
        Dl, d, Dr = A.shape
        Dlb, _, Drb = Ab.shape
        W = GRin.shape[0]
        Wout = H.Wl if H is not None else W
        y = self.empty(Wout, Dl, Dlb) if out is None else out
        check(self.lib.mpsk_transfer_right(self.ctx, H.handle if H is not None else None, W, d, Dl, Dr, Dlb, Drb,
                                           A.ptr, Ab.ptr, GRin.ptr, y.ptr), "mpsk_transfer_right")
        return y

    def regularize(self, v: DTensor, lvec: DTensor, rvec: DTensor):
        W, D1, D2 = v.shape
        check(self.lib.mpsk_regularize(self.ctx, W, D1, D2, v.ptr, lvec.ptr, rvec.ptr), "mpsk_regularize")
        return v

    def gemm(self, A: DTensor, B: DTensor, transA=False, transB=False, alpha=1.0, beta=0.0, out: DTensor = None,
             m=None, n=None, k=None, lda=None, ldb=None, ldc=None):
        """C = alpha op(A) op(B) + beta C on 2-D column-major views (leading dims default to rows)."""
        ar, ac = A.shape
        br, bc = B.shape
        M = (ac if transA else ar) if m is None else m
        K = (ar if transA else ac) if k is None else k
        N = (br if transB else bc) if n is None else n
        c = self.empty(M, N) if out is None else out
        check(self.lib.mpsk_gemm(self.ctx, int(transA), int(transB), M, N, K, float(alpha), A.ptr,
                                 ar if lda is None else lda, B.ptr, br if ldb is None else ldb, float(beta),
                                 c.ptr, M if ldc is None else ldc), "mpsk_gemm")
        return c

    def gemm_raw(self, transA, transB, M, N, K, alpha, a_ptr, lda, b_ptr, ldb, beta, c_ptr, ldc):
        check(self.lib.mpsk_gemm(self.ctx, int(transA), int(transB), M, N, K, float(alpha), a_ptr, lda, b_ptr,
                                 ldb, float(beta), c_ptr, ldc), "mpsk_gemm")

    def triu_(self, M: DTensor):
        """zero the strictly lower triangle of a column-major matrix in place (torch view of the buffer: a fill, no arithmetic)."""
        torch = _torch()
        r, c = M.shape
        v = torch.as_strided(M.buf, (r, c), (1, r))
        v.copy_(torch.triu(v))
        return M

    def tril_(self, M: DTensor):
        torch = _torch()
        r, c = M.shape
        v = torch.as_strided(M.buf, (r, c), (1, r))
        v.copy_(torch.tril(v))
        return M

    def copy2d(self, rows, cols, src_ptr, lds, dst_ptr, ldd):
        check(self.lib.mpsk_copy2d(self.ctx, rows, cols, src_ptr, lds, dst_ptr, ldd), "mpsk_copy2d")

    # ---- gauge -------------------------------------------------------------------------------
    def qrpos(self, A: DTensor):
        m, n = A.shape
        k = min(m, n)
        Q, R = self.empty(m, k), self.empty(k, n)
        check(self.lib.mpsk_qrpos(self.ctx, m, n, A.ptr, m, Q.ptr, m, R.ptr, k), "mpsk_qrpos")
        return Q, R

    def cx_embed_raw(self, m, n, h_ptr, ldh, e_ptr, lde):
        """embedded (2m x 2n) <- interleaved complex (2m x n), one launch (mpsk_cx_embed; leading dimensions in doubles)"""
        check(self.lib.mpsk_cx_embed(self.ctx, m, n, h_ptr, ldh, e_ptr, lde), "mpsk_cx_embed")

    def cx_half_raw(self, m, n, e_ptr, lde, h_ptr, ldh):
        """interleaved complex (2m x n) <- structured part of the embedded (2m x 2n), one launch (mpsk_cx_half)"""
        check(self.lib.mpsk_cx_half(self.ctx, m, n, e_ptr, lde, h_ptr, ldh), "mpsk_cx_half")

    def gemm_c(self, A: DTensor, B: DTensor, transA=False, transB=False, alpha=1.0, beta=0.0, out: DTensor = None):
        """C = alpha op(A) op(B) + beta C on interleaved complex128 matrices (shape (2 rows, cols); op = conjugate transpose;
        mpsk_gemm under MPSK_C128: the small gauge products AC = AL*C, AL = Q_AC*Q_C' of a complex host)."""
        ar, ac = A.shape[0] // 2, A.shape[1]
        br, bc = B.shape[0] // 2, B.shape[1]
        M, K = (ac, ar) if transA else (ar, ac)
        K2, N = (bc, br) if transB else (br, bc)
        assert K == K2
        out = self.empty(2 * M, N) if out is None else out
        self._set_dtype(True)
        try:
            check(self.lib.mpsk_gemm(self.ctx, int(transA), int(transB), M, N, K, float(alpha), A.ptr, ar, B.ptr, br,
                                     float(beta), out.ptr, M), "mpsk_gemm (C128)")
        finally:
            self._set_dtype(False)
        return out

    def qrpos_c(self, A: DTensor):
        """QRpos of an interleaved complex128 matrix (upload_c layout: shape (2 m, n) = complex m x n): returns Q (2 m, n) and
        R (2 n, n), complex upper triangular with a real positive diagonal (mpsk_qrpos under MPSK_C128)."""
        m2, n = A.shape
        m = m2 // 2
        assert m2 == 2 * m and m >= n
        Q, R = self.empty(m2, n), self.empty(2 * n, n)
        self._set_dtype(True)
        try:
            check(self.lib.mpsk_qrpos(self.ctx, m, n, A.ptr, m, Q.ptr, m, R.ptr, n), "mpsk_qrpos (C128)")
        finally:
            self._set_dtype(False)
        return Q, R

    def lqpos_c(self, A: DTensor):
        """LQpos of an interleaved complex128 matrix (shape (2 m, n), m <= n): L (2 m, m) lower triangular with a real positive
        diagonal, Q (2 m, n) with orthonormal rows (mpsk_lqpos under MPSK_C128)."""
        m2, n = A.shape
        m = m2 // 2
        assert m2 == 2 * m and m <= n
        L, Q = self.empty(m2, m), self.empty(m2, n)
        self._set_dtype(True)
        try:
            check(self.lib.mpsk_lqpos(self.ctx, m, n, A.ptr, m, L.ptr, m, Q.ptr, m), "mpsk_lqpos (C128)")
        finally:
            self._set_dtype(False)
        return L, Q

    def tsplit_c(self, theta: DTensor, max_keep=0):
        """Two-site split of an interleaved complex128 tensor (shape (2 m, n) = complex m x n; mpsk_tsplit under MPSK_C128,
        truncdim scheme): al (2 m, k), c (2 k, k) lower triangular, ar (2 k, n), the k kept singular values, discarded norm."""
        m2, n = theta.shape
        m = m2 // 2
        kf = min(m, n)
        AL, Cm, AR, S = self.empty(m2, kf), self.empty(2 * kf, kf), self.empty(2 * kf, n), self.empty(kf)
        kept, disc = C.c_int(0), C.c_double(0.0)
        self._set_dtype(True)
        try:
            check(self.lib.mpsk_tsplit(self.ctx, m, n, theta.ptr, m, int(max_keep), 0.0, AL.ptr, m, Cm.ptr, kf, AR.ptr, kf,
                                       S.ptr, C.byref(kept), C.byref(disc)), "mpsk_tsplit (C128)")
        finally:
            self._set_dtype(False)
        k = kept.value
        c, ar = self.empty(2 * k, k), self.empty(2 * k, n)
        self.copy2d(2 * k, k, Cm.ptr, 2 * kf, c.ptr, 2 * k)
        self.copy2d(2 * k, n, AR.ptr, 2 * kf, ar.ptr, 2 * k)
        return DTensor(AL.buf, (m2, k)), c, ar, self.download(DTensor(S.buf, (k,))), disc.value

    def qrpos2(self, A1: DTensor, A2: DTensor):
        """two QRpos of equal shape in flight together (two streams inside the ctx)."""
        m, n = A1.shape
        assert A2.shape == (m, n) and m >= n
        Q1, R1, Q2, R2 = self.empty(m, n), self.empty(n, n), self.empty(m, n), self.empty(n, n)
        check(self.lib.mpsk_qrpos2(self.ctx, m, n, A1.ptr, m, Q1.ptr, m, R1.ptr, n, A2.ptr, m, Q2.ptr, m, R2.ptr, n),
              "mpsk_qrpos2")
        return Q1, R1, Q2, R2

    def qr_defer(self):
        """the next qrpos2 / lqpos returns once enqueued; finish it with qr_commit (include/mpsk.h)."""
        check(self.lib.mpsk_ctx_qr_defer(self.ctx), "mpsk_ctx_qr_defer")

    def side_mark(self):
        check(self.lib.mpsk_ctx_side_mark(self.ctx), "mpsk_ctx_side_mark")

    def side_begin(self):
        """route the following calls to the ctx's second stream, ordered after side_mark() only (include/mpsk.h)."""
        check(self.lib.mpsk_ctx_side_begin(self.ctx), "mpsk_ctx_side_begin")

    def side_end(self):
        check(self.lib.mpsk_ctx_side_end(self.ctx), "mpsk_ctx_side_end")

    def qr_commit(self) -> int:
        r = C.c_int(0)
        check(self.lib.mpsk_qr_commit(self.ctx, C.byref(r)), "mpsk_qr_commit")
        return r.value

    def qrlq_pair(self, A1: DTensor, A2: DTensor):
        """QRpos of A1 (m x n) and LQpos of A2 (n x m) in flight together -> (Q1, R1, L2, Q2)."""
        m, n = A1.shape
        assert A2.shape == (n, m) and m >= n
        Q1, R1, L2, Q2 = self.empty(m, n), self.empty(n, n), self.empty(n, n), self.empty(n, m)
        check(self.lib.mpsk_qrlq_pair(self.ctx, m, n, A1.ptr, m, Q1.ptr, m, R1.ptr, n, A2.ptr, n, L2.ptr, n, Q2.ptr, n),
              "mpsk_qrlq_pair")
        return Q1, R1, L2, Q2

    def lqpos(self, A: DTensor):
        m, n = A.shape
        k = min(m, n)
        L, Q = self.empty(m, k), self.empty(k, n)
        check(self.lib.mpsk_lqpos(self.ctx, m, n, A.ptr, m, L.ptr, m, Q.ptr, k), "mpsk_lqpos")
        return L, Q

    def tsvd(self, theta: DTensor, max_keep=0, trunc_err=0.0):
        m, n = theta.shape
        kmax = min(m, n)
        U, S, Vh = self.empty(m, kmax), self.empty(kmax), self.empty(kmax, n)
        kept, disc = C.c_int(0), C.c_double(0.0)
        check(self.lib.mpsk_tsvd(self.ctx, m, n, theta.ptr, m, U.ptr, m, S.ptr, Vh.ptr, kmax, int(max_keep),
                                 float(trunc_err), C.byref(kept), C.byref(disc)), "mpsk_tsvd")
        return U, S, Vh, kept.value, disc.value

    def tsplit(self, theta: DTensor, max_keep=0, trunc_err=0.0):
        """Truncated two-site split theta (m x n) ~ al (m x k) . c (k x k) . ar (k x n)  (mpsk_tsplit: V-free Jacobi;
        small tensors go through mpsk_tsvd with c = diag(S)).  Returns (al, c, ar, S[:k] on the host, disc)."""
        m, n = theta.shape
        kmax = min(m, n)
        if kmax <= 64:
            U, S, Vh, k, disc = self.tsvd(theta, max_keep=max_keep, trunc_err=trunc_err)
            s = self.download(DTensor(S.buf, (k,)))
            al = self.empty(m, k)
            self.copy2d(m, k, U.ptr, m, al.ptr, m)
            ar = self.empty(k, n)
            self.copy2d(k, n, Vh.ptr, kmax, ar.ptr, k)
            if k > 0 and s[k - 1] <= 1e-14 * s[0]:
                # kept singular values at the rounding floor (rank-deficient theta with truncdim > rank): the one-sided
                # Jacobi leaves (numerically) zero vectors there, LAPACK returns an orthonormal completion.  QRpos /
                # LQpos (Householder for <= 64 columns) reproduce the well-defined columns and complete the rest,
                # so al / ar stay isometries, which the lazy gauge and FinEnv assume.
                al, _ = self.qrpos(al)
                _, ar = self.lqpos(ar)
            return al, self.upload(np.diag(s)), ar, s, disc
        AL, Cm, AR, S = self.empty(m, kmax), self.empty(kmax, kmax), self.empty(kmax, n), self.empty(kmax)
        kept, disc = C.c_int(0), C.c_double(0.0)
        check(self.lib.mpsk_tsplit(self.ctx, m, n, theta.ptr, m, int(max_keep), float(trunc_err), AL.ptr, m, Cm.ptr, kmax,
                                   AR.ptr, kmax, S.ptr, C.byref(kept), C.byref(disc)), "mpsk_tsplit")
        k = kept.value
        al = DTensor(AL.buf, (m, k))                       # leading k columns, ld = m
        c = self.empty(k, k)
        self.copy2d(k, k, Cm.ptr, kmax, c.ptr, k)
        ar = self.empty(k, n)
        self.copy2d(k, n, AR.ptr, kmax, ar.ptr, k)
        return al, c, ar, self.download(DTensor(S.buf, (k,))), disc.value

    # ---- vectors -----------------------------------------------------------------------------
    def _ptrs(self, xs):
        arr = (C.c_void_p * len(xs))(*[x.ptr for x in xs])
        return arr

    def dot(self, x: DTensor, y: DTensor):
        out = C.c_double()
        check(self.lib.mpsk_vdot(self.ctx, x.size, x.ptr, y.ptr, C.byref(out)), "mpsk_vdot")
        return out.value

    def norm(self, x: DTensor):
        out = C.c_double()
        check(self.lib.mpsk_vnrm2(self.ctx, x.size, x.ptr, C.byref(out)), "mpsk_vnrm2")
        return out.value

    def axpby(self, alpha, x: DTensor, beta, y: DTensor):
        check(self.lib.mpsk_vaxpby(self.ctx, x.size, float(alpha), x.ptr, float(beta), y.ptr), "mpsk_vaxpby")
        return y

    def times_i(self, x: DTensor, out: DTensor = None):
        """embedded complex tensors (cplx.py): out = i * x  (first dimension = interleaved re/im rows)."""
        y = self.empty(*x.shape) if out is None else out
        check(self.lib.mpsk_vtimes_i(self.ctx, x.size, x.ptr, y.ptr), "mpsk_vtimes_i")
        return y

    def scal(self, alpha, x: DTensor):
        check(self.lib.mpsk_vscal(self.ctx, x.size, float(alpha), x.ptr), "mpsk_vscal")
        return x

    def multidot(self, xs, y: DTensor):
        out = (C.c_double * len(xs))()
        check(self.lib.mpsk_vmultidot(self.ctx, y.size, len(xs), self._ptrs(xs), y.ptr, out), "mpsk_vmultidot")
        return np.array(out[:])

    def gs_step(self, xs, y: DTensor):
        out = (C.c_double * len(xs))()
        check(self.lib.mpsk_vgs_step(self.ctx, y.size, len(xs), self._ptrs(xs), y.ptr, out), "mpsk_vgs_step")
        return np.array(out[:])

    def orth_step(self, xs, y: DTensor):
        """CGS2 of y against xs, then y <- y/||y||; ONE host sync.  Returns (h[k], beta)."""
        out = (C.c_double * len(xs))()
        beta = C.c_double()
        check(self.lib.mpsk_vorth_step(self.ctx, y.size, len(xs), self._ptrs(xs), y.ptr, out, C.byref(beta)),
              "mpsk_vorth_step")
        return np.array(out[:]), beta.value

    class _OrthHandle:
        """result of orth_step_async: .result() waits for the step's scalars only (an event), not for the stream, and
        hands the (device, pinned host) buffer pair back to the backend's free list"""
        __slots__ = ("be", "slot", "event", "k")

        def __init__(self, be, slot, event, k):
            self.be, self.slot, self.event, self.k = be, slot, event, k

        def result(self):
            self.event.synchronize()
            k = self.k
            t = self.slot[1].numpy()
            h = (t[:k] + t[k:2 * k]).copy()
            beta = float(np.sqrt(max(t[2 * k], 0.0)))
            self.be._orth_free.append(self.slot)
            self.slot = None
            return h, beta

        def __del__(self):                      # a speculative step that was dropped: the buffers go back as well
            if self.slot is not None:           # (stream order: the dropped copy completes before any later use of them)
                self.be._orth_free.append(self.slot)

    ORTH_SLOT = 544                             # 2 * 271 + 2 doubles

    def orth_step_async(self, xs, y: DTensor):
        """orth_step whose scalars travel to the host asynchronously: the call returns once the CGS2 / normalisation
        kernels and the device-to-host copy of the 2k+1 scalars are enqueued.  A tolerance-mode Krylov loop enqueues the
        NEXT matvec before it reads the handle, so the GPU works while the host takes its per-step decision (the stream used
        to drain at every step: 36 % idle in the VUMPS run of BASELINE config 3, profiles/r03_small_D_kernel_stats.log).
        Every handle owns its buffers until it is read or dropped: solvers nest (the quasiparticle effective Hamiltonian
        runs GMRES solves inside the matvec of an outer eigensolver that has a handle in flight)."""
        torch = _torch()
        k = len(xs)
        n = 2 * k + 1
        if n > self.ORTH_SLOT:
            raise MpskError("orth_step_async: more than 271 basis vectors")
        free = getattr(self, "_orth_free", None)
        if free is None:
            free = self._orth_free = []
        slot = free.pop() if free else (torch.empty(self.ORTH_SLOT, dtype=torch.float64, device=self.device),
                                        torch.empty(self.ORTH_SLOT, dtype=torch.float64).pin_memory())
        check(self.lib.mpsk_vorth_step_dev(self.ctx, y.size, k, self._ptrs(xs), y.ptr, slot[0].data_ptr()), "mpsk_vorth_step_dev")
        with torch.cuda.stream(self.torch_stream):
            slot[1][:n].copy_(slot[0][:n], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.torch_stream)
        return Backend._OrthHandle(self, slot, ev, k)

    def orth_step_dev(self, xs, y: DTensor, slot: DTensor, offset: int):
        """orth_step without the host sync: the 2k+1 scalars go to slot[offset : offset + 2k + 1] on the device."""
        check(self.lib.mpsk_vorth_step_dev(self.ctx, y.size, len(xs), self._ptrs(xs), y.ptr, slot.ptr + 8 * offset),
              "mpsk_vorth_step_dev")

    def normalize_dev(self, x: DTensor, out: DTensor = None, slot: DTensor = None, offset: int = 0):
        """out = x / |x| (default in place) with no host sync; |x|^2 goes to slot[offset] (device) when given."""
        y = x if out is None else out
        check(self.lib.mpsk_vnormalize_dev(self.ctx, x.size, x.ptr, y.ptr, None if slot is None else slot.ptr + 8 * offset),
              "mpsk_vnormalize_dev")
        return y

    def nrm2_dev(self, x: DTensor, slot: DTensor, offset: int = 0):
        """slot[offset] = |x|^2 on the device, no host sync."""
        check(self.lib.mpsk_vnrm2_dev(self.ctx, x.size, x.ptr, slot.ptr + 8 * offset), "mpsk_vnrm2_dev")

    RITZ_BUF = 40   # doubles: coefficients [0:32], info (lambda, residual estimate, effective m) [32:35]

    def ritz_dev(self, m: int, stride: int, slot: DTensor, buf: DTensor):
        """Ritz coefficients of a fixed-budget solve on the device (mpsk_vritz_dev): buf[0:m], buf[32:35] = info."""
        check(self.lib.mpsk_vritz_dev(self.ctx, int(m), int(stride), slot.ptr, buf.ptr, buf.ptr + 8 * 32), "mpsk_vritz_dev")

    def lincomb_dev(self, xs, coef: DTensor, out: DTensor = None):
        y = self.empty(xs[0].shape) if out is None else out
        check(self.lib.mpsk_vlincomb_dev(self.ctx, y.size, len(xs), self._ptrs(xs), coef.ptr, y.ptr), "mpsk_vlincomb_dev")
        return y

    def multilincomb(self, xs, S, outs):
        """outs[j] = sum_i S[i, j] xs[i] in one pass over the vectors (mpsk_vmultilincomb: the basis rotation of a thick
        restart); len(xs), len(outs) <= 32, outs disjoint from xs."""
        S = np.asarray(S, dtype=float)
        k, m = len(xs), len(outs)
        assert S.shape == (k, m)
        cf = (C.c_double * (k * m))(*S.T.reshape(-1))                # column j of S contiguous: coefs[i + k j]
        check(self.lib.mpsk_vmultilincomb(self.ctx, xs[0].size, k, self._ptrs(xs), m, self._ptrs(outs), cf),
              "mpsk_vmultilincomb")
        return outs

    def lincomb(self, xs, coefs, out: DTensor = None):
        y = self.empty(xs[0].shape) if out is None else out
        cf = (C.c_double * len(xs))(*[float(c) for c in coefs])
        check(self.lib.mpsk_vlincomb(self.ctx, y.size, len(xs), self._ptrs(xs), cf, y.ptr), "mpsk_vlincomb")
        return y


class PreparedHAC:
    """Owner of an mpsk_hac handle (include/mpsk.h): `apply(x, out, nblk)` is one matvec."""

    def __init__(self, be: Backend, H, GL: DTensor, GR: DTensor):
        Wl, Dlo, Dl = GL.shape
        Wr, Dr2, Dr = GR.shape
        self.cplx = bool(getattr(H, "cplx", False))
        m = 2 if self.cplx else 1                                   # interleaved complex: doubled first dimensions
        Dlo //= m
        assert Wl == H.Wl and Wr == H.Wr and Dr2 == m * Dr, (GL.shape, GR.shape)
        self.be, self.H, self.GL, self.GR = be, H, GL, GR           # references keep the operands alive
        self.Dlo, self.Dl, self.Dr, self.d = Dlo, Dl, Dr, H.d
        h = C.c_void_p()
        check(be.lib.mpsk_hac_create(be.ctx, H.handle, Dlo, Dl, Dr, GL.ptr, GR.ptr, C.byref(h)), "mpsk_hac_create")
        self.handle = h

    def info(self):
        mode, ns = C.c_int(), C.c_int()
        check(self.be.lib.mpsk_hac_info(self.handle, C.byref(mode), C.byref(ns)), "mpsk_hac_info")
        return {"mode": mode.value, "combined_slabs": ns.value}

    def apply(self, x: DTensor, out: DTensor = None, nblk=1):
        m = 2 if self.cplx else 1
        assert x.shape == (m * self.Dl, self.d, self.Dr), (x.shape, (m * self.Dl, self.d, self.Dr))
        y = self.be.empty(m * self.Dlo, self.d, self.Dr) if out is None else out
        check(self.be.lib.mpsk_hac_apply(self.handle, x.ptr, int(nblk), y.ptr), "mpsk_hac_apply")
        return y

    def eigsolve_fixed(self, x0: DTensor, m: int, vecs, scal: DTensor, out: DTensor, first_image: DTensor = None):
        """mpsk_hac_eigsolve_fixed: the whole fixed-budget solve (m steps) in one call; vecs = m + 2 work vectors."""
        assert not self.cplx and len(vecs) >= m + 2 and scal.size >= m * (2 * m + 1) + 40
        ptrs = self.be._ptrs(vecs[:m + 2])
        check(self.be.lib.mpsk_hac_eigsolve_fixed(self.handle, x0.ptr, int(m), ptrs, scal.ptr, out.ptr,
                                                  None if first_image is None else first_image.ptr), "mpsk_hac_eigsolve_fixed")
        return out

    def close(self):
        if getattr(self, "handle", None) and self.be.ctx:
            self.be.lib.mpsk_hac_destroy(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceMPOSlice:
    """Device-side SparseMPOSlice (sparseslice.jl:13-27): keeps the host block table for the
    host-side logic (keys / isscal / contains) and an mpsk_mposlice handle for the kernels."""

    def __init__(self, be: Backend, odim, d, chil, chir, blocks, cplx=False):
        self.be = be
        self.cplx = bool(cplx)
        self.odim, self.d = int(odim), int(d)
        self.chil, self.chir = [int(c) for c in chil], [int(c) for c in chir]
        self.Wl, self.Wr = sum(self.chil), sum(self.chir)
        self.blocks = dict(blocks)
        n = self.odim
        kind = (C.c_int32 * (n * n))()
        scal = (C.c_double * ((2 if self.cplx else 1) * n * n))()
        ptrs = (C.c_void_p * (n * n))()
        keep = []
        for (i, j), v in self.blocks.items():
            if np.isscalar(v):
                if not self.cplx and np.iscomplexobj(v) and complex(v).imag != 0:
                    raise MpskError("complex MPO entries need an MPSK_C128 slice (mposlice(..., cplx=True))")
                if v == 0:
                    continue
                kind[i + n * j] = 1
                if self.cplx:
                    scal[2 * (i + n * j)], scal[2 * (i + n * j) + 1] = float(np.real(v)), float(np.imag(v))
                else:
                    scal[i + n * j] = float(np.real(v))
            else:
                a = np.asarray(v)
                if np.iscomplexobj(a) and not self.cplx:
                    if np.abs(a.imag).max() > 0:
                        raise MpskError("complex MPO entries need an MPSK_C128 slice (mposlice(..., cplx=True))")
                    a = a.real
                assert a.shape == (self.chil[i], d, d, self.chir[j]), (a.shape, i, j)
                if self.cplx:
                    flat = np.ravel(np.asarray(a, dtype=np.complex128), order="F").view(np.float64).copy()
                else:
                    flat = np.ravel(np.asfortranarray(a, dtype=np.float64), order="F").copy()
                keep.append(flat)
                kind[i + n * j] = 2
                ptrs[i + n * j] = flat.ctypes.data
        h = C.c_void_p()
        cl = (C.c_int32 * n)(*self.chil)
        cr = (C.c_int32 * n)(*self.chir)
        check(be.lib.mpsk_mposlice_create(be.ctx, 1 if self.cplx else 0, n, cl, cr, self.d, kind, scal, ptrs, C.byref(h)),
              "mpsk_mposlice_create")
        self.handle = h

    def __del__(self):
        try:
            if self.handle and self.be.ctx:
                self.be.lib.mpsk_mposlice_destroy(self.handle)
        except Exception:
            pass
        self.handle = None

    # host-side helpers mirroring sparseslice.jl:74-106
    def keys(self):
        return sorted(self.blocks.keys(), key=lambda t: (t[1], t[0]))

    def contains(self, i, j):
        return (i, j) in self.blocks and not (np.isscalar(self.blocks[(i, j)]) and self.blocks[(i, j)] == 0)

    def isscal(self, i, j):
        return self.contains(i, j) and np.isscalar(self.blocks[(i, j)])


_default = {}


def default_backend(device=None):
    torch = _torch()
    if device is None:
        device = torch.cuda.current_device() if torch.cuda.is_available() else 0
    if device not in _default:
        _default[device] = Backend(device)
    return _default[device]
