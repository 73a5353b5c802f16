"""TEST INFRASTRUCTURE: a host stand-in for `mpskit_jl_amd.backend.Backend`.

It lets the `-m "not gpu"` suite exercise the PRODUCT's host logic (lazy-gauge FiniteMPS, FinEnv
invalidation, Krylov loops, DMRG driver, bond-sharding plumbing) without a GPU: same method
names / argument conventions as Backend, DTensor buffers are CPU torch tensors holding the same
column-major data, arithmetic is done by NumPy / the oracle.  It is never imported by the
product package (which has no CPU fallback)."""
import ctypes as C

import numpy as np
import torch

import mpskit_oracle as mo
from mpskit_jl_amd.backend import DTensor


def _view(ptr, n):
    return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_double)), shape=(n,))


class HostSlice:
    def __init__(self, odim, d, chil, chir, blocks):
        self.odim, self.d = odim, d
        self.chil, self.chir = list(chil), list(chir)
        self.Wl, self.Wr = sum(chil), sum(chir)
        self.blocks = dict(blocks)
        self.oracle = mo.SparseMPOSlice(odim, d, chil, chir, {k: v for k, v in blocks.items()})
        self.handle = None

    def keys(self):
        return self.oracle.keys()

    def contains(self, i, j):
        return self.oracle.contains(i, j)

    def isscal(self, i, j):
        return self.oracle.isscal(i, j)


class CpuBackend:
    device = torch.device("cpu")

    def __init__(self):
        self.calls = {}
        self._qr = {"cholqr3": 0, "householder": 0, "fallback": 0, "robust": 0}

    def _count(self, name):
        self.calls[name] = self.calls.get(name, 0) + 1

    # ---- memory ----
    def empty(self, *shape):
        if len(shape) == 1 and isinstance(shape[0], (tuple, list)):
            shape = tuple(shape[0])
        n = int(np.prod(shape)) if len(shape) else 1
        return DTensor(torch.zeros(max(n, 1), dtype=torch.float64), shape)

    zeros = empty

    def upload(self, a, shape=None):
        a = np.asarray(a, dtype=np.float64)
        flat = np.ravel(a, order="F").copy()
        if flat.size == 0:
            flat = np.zeros(1)
        return DTensor(torch.from_numpy(flat), a.shape if shape is None else shape)

    def download(self, t):
        return t.buf[: t.size].numpy().reshape(t.shape, order="F").copy()

    def _set(self, t, arr):
        t.buf[: t.size] = torch.from_numpy(np.ravel(np.asarray(arr, dtype=np.float64), order="F").copy())
        return t

    def upload_env(self, blocks):
        slabs = [np.asarray(b)[:, k, :] for b in blocks for k in range(np.asarray(b).shape[1])]
        flat = np.concatenate([np.ravel(s, order="F") for s in slabs])
        return DTensor(torch.from_numpy(flat.copy()), (len(slabs),) + slabs[0].shape)

    def _env(self, t, chis):
        W, Db, Dk = t.shape
        flat = t.buf[: t.size].numpy()
        slabs = [flat[w * Db * Dk:(w + 1) * Db * Dk].reshape((Db, Dk), order="F") for w in range(W)]
        out, o = [], 0
        for chi in chis:
            out.append(np.stack(slabs[o:o + chi], axis=1))
            o += chi
        return out

    download_env = _env

    def _put_env(self, blocks, out=None):
        t = self.upload_env(blocks)
        if out is not None:
            out.buf[: t.size] = t.buf[: t.size]
            return out
        return t

    def copy(self, t):
        return DTensor(t.buf.clone(), t.shape)

    def synchronize(self):
        pass

    def mposlice(self, odim, d, chil, chir, blocks):
        return HostSlice(odim, d, chil, chir, blocks)

    # ---- operators (oracle arithmetic) ----
    def dAC(self, H, GL, GR, x, out=None):
        self._count("dAC")
        gl = self._env(GL, H.chil)
        y = mo.dAC(self.download(x), H.oracle, gl, self._env(GR, H.chir))
        if y is None:
            y = np.zeros((GL.shape[1],) + x.shape[1:])
        o = self.empty(GL.shape[1], x.shape[1], x.shape[2]) if out is None else out
        return self._set(o, y)

    def dAC_blocked(self, H, GLrows, GR, xb, nblk, out=None):
        """x in nblk row blocks (dist.to_blocked) -> un-block on the host, then the plain dAC on the local rows."""
        self._count("dAC_blocked")
        Dl, d, Dr = xb.shape
        n = Dl // nblk
        flat = xb.buf[: xb.size].numpy()
        x = np.concatenate([flat[q * n * d * Dr:(q + 1) * n * d * Dr].reshape((n, d, Dr), order="F") for q in range(nblk)], axis=0)
        y = mo.dAC(x, H.oracle, self._env(GLrows, H.chil), self._env(GR, H.chir))
        if y is None:
            y = np.zeros((GLrows.shape[1], d, Dr))
        o = self.empty(GLrows.shape[1], d, Dr) if out is None else out
        return self._set(o, y)

    def dAC2(self, H1, H2, GL, GR, x2, out=None):
        self._count("dAC2")
        y = mo.dAC2(self.download(x2), H1.oracle, H2.oracle, self._env(GL, H1.chil), self._env(GR, H2.chir))
        o = self.empty(GL.shape[1], x2.shape[1], GR.shape[2], x2.shape[3]) if out is None else out
        return self._set(o, y)

    def tsplit(self, theta, max_keep=0, trunc_err=0.0):
        """Backend.tsplit: (al, c, ar, S[:k], disc) with c = diag(S) (the GPU path returns a triangular c for large theta)."""
        self._count("tsplit")
        U, S, Vh, k, disc = self.tsvd(theta, max_keep=max_keep, trunc_err=trunc_err)
        u, s, vh = self.download(U), self.download(S).reshape(-1), self.download(Vh)
        return self.upload(u[:, :k]), self.upload(np.diag(s[:k])), self.upload(vh[:k, :]), s[:k].copy(), disc

    def dC(self, GL, GR, c, out=None):
        self._count("dC")
        W = GL.shape[0]
        y = mo.dC(self.download(c), self._env(GL, [1] * W), self._env(GR, [1] * W))
        o = self.empty(GL.shape[1], c.shape[1]) if out is None else out
        return self._set(o, y)

    def transfer_left(self, H, GLin, A, Ab, out=None):
        self._count("transfer_left")
        a, ab = self.download(A), self.download(Ab)
        if H is None:
            W = GLin.shape[0]
            res = [mo.transfer_left_block(v, None, a, ab) for v in self._env(GLin, [1] * W)]
        else:
            res = mo.transfer_left(self._env(GLin, H.chil), H.oracle, a, ab)
        return self._put_env(res, out)

    def transfer_right(self, H, GRin, A, Ab, out=None):
        self._count("transfer_right")
        a, ab = self.download(A), self.download(Ab)
        if H is None:
            W = GRin.shape[0]
            res = [mo.transfer_right_block(v, None, a, ab) for v in self._env(GRin, [1] * W)]
        else:
            res = mo.transfer_right(self._env(GRin, H.chir), H.oracle, a, ab)
        return self._put_env(res, out)

    def regularize(self, v, lvec, rvec):
        W = v.shape[0]
        l, r = self.download(lvec), self.download(rvec)
        res = [mo.regularize_env(e, l, r) for e in self._env(v, [1] * W)]
        self._put_env(res, v)
        return v

    def gemm(self, A, B, transA=False, transB=False, alpha=1.0, beta=0.0, out=None, **kw):
        a, b = self.download(A), self.download(B)
        r = alpha * (a.T if transA else a) @ (b.T if transB else b)
        if out is None:
            return self.upload(r)
        if beta != 0.0:
            r = r + beta * self.download(out.reshape(r.shape))
        return self._set(out, r)

    def gemm_raw(self, tA, tB, M, N, K, alpha, a_ptr, lda, b_ptr, ldb, beta, c_ptr, ldc):
        ar, ac = (K, M) if tA else (M, K)
        br, bc = (N, K) if tB else (K, N)
        a = _view(a_ptr, lda * ac).reshape((lda, ac), order="F")[:ar]
        b = _view(b_ptr, ldb * bc).reshape((ldb, bc), order="F")[:br]
        c = _view(c_ptr, ldc * N).reshape((ldc, N), order="F")
        r = alpha * (a.T if tA else a) @ (b.T if tB else b)
        c[:M] = r + (beta * c[:M] if beta != 0.0 else 0.0)

    def copy2d(self, rows, cols, src_ptr, lds, dst_ptr, ldd):
        s = _view(src_ptr, lds * cols).reshape((lds, cols), order="F")
        d = _view(dst_ptr, ldd * cols).reshape((ldd, cols), order="F")
        d[:rows] = s[:rows]

    def qrpos(self, A):
        self._count("qrpos")
        q, r = mo.qrpos(self.download(A))
        return self.upload(q), self.upload(r)

    def qrpos2(self, A1, A2):
        self._count("qrpos2")
        q1, r1 = self.qrpos(A1)
        q2, r2 = self.qrpos(A2)
        return q1, r1, q2, r2

    def qrlq_pair(self, A1, A2):
        self._count("qrlq_pair")
        q1, r1 = self.qrpos(A1)
        l2, q2 = self.lqpos(A2)
        return q1, r1, l2, q2

    def lqpos(self, A):
        self._count("lqpos")
        l, q = mo.lqpos(self.download(A))
        return self.upload(l), self.upload(q)

    def tsvd(self, theta, max_keep=0, trunc_err=0.0):
        """mpsk_tsvd: full thin factors + the number kept under truncdim / truncerr and the discarded 2-norm."""
        self._count("tsvd")
        a = self.download(theta)
        U, S, Vh = np.linalg.svd(a, full_matrices=False)
        k = len(S) if not max_keep else min(len(S), int(max_keep))
        if trunc_err > 0:
            while k > 1 and np.sqrt(np.sum(S[k - 1:] ** 2)) <= trunc_err:
                k -= 1
        return self.upload(U), self.upload(S), self.upload(Vh), k, float(np.sqrt(np.sum(S[k:] ** 2)))

    def qr_stats(self):
        return dict(self._qr)

    # ---- vectors ----
    def _v(self, x):
        return x.buf[: x.size].numpy()

    def dot(self, x, y):
        return float(self._v(x) @ self._v(y))

    def norm(self, x):
        return float(np.linalg.norm(self._v(x)))

    def axpby(self, alpha, x, beta, y):
        v = alpha * self._v(x) + (beta * self._v(y) if beta != 0.0 else 0.0)
        y.buf[: y.size] = torch.from_numpy(np.ascontiguousarray(v))
        return y

    def times_i(self, x, out=None):
        v = self._v(x).reshape(-1, 2)
        y = self.empty(*x.shape) if out is None else out
        self._v(y)[:] = np.stack([-v[:, 1], v[:, 0]], axis=1).reshape(-1)
        return y

    def scal(self, alpha, x):
        x.buf[: x.size] *= alpha
        return x

    def multidot(self, xs, y):
        return np.array([self.dot(x, y) for x in xs])

    def gs_step(self, xs, y):
        h = self.multidot(xs, y)
        v = self._v(y) - sum(c * self._v(x) for c, x in zip(h, xs))
        y.buf[: y.size] = torch.from_numpy(np.ascontiguousarray(v))
        return h

    def orth_step(self, xs, y):
        h = self.gs_step(xs, y)
        h = h + self.gs_step(xs, y)
        beta = self.norm(y)
        if beta > 0:
            self.scal(1.0 / beta, y)
        return h, beta

    def orth_step_dev(self, xs, y, slot, offset):
        k = len(xs)
        h1 = self.gs_step(xs, y)
        h2 = self.gs_step(xs, y)
        n2 = self.norm(y) ** 2
        if n2 > 0:
            self.scal(1.0 / np.sqrt(n2), y)
        buf = slot.buf.numpy()
        buf[offset:offset + k] = h1
        buf[offset + k:offset + 2 * k] = h2
        buf[offset + 2 * k] = n2

    def normalize_dev(self, x, out=None, slot=None, offset=0):
        y = x if out is None else out
        n2 = float(self._v(x) @ self._v(x))
        v = self._v(x) * (1.0 / np.sqrt(n2) if n2 > 0 else 0.0)
        y.buf[: y.size] = torch.from_numpy(np.ascontiguousarray(v))
        if slot is not None:
            slot.buf.numpy()[offset] = n2
        return y

    def nrm2_dev(self, x, slot, offset=0):
        slot.buf.numpy()[offset] = float(self._v(x) @ self._v(x))

    def qr_defer(self):
        pass

    def side_mark(self):
        pass

    def side_begin(self):
        pass

    def side_end(self):
        pass

    def qr_commit(self):
        return 0

    RITZ_BUF = 40

    def ritz_dev(self, m, stride, slot, buf):
        co = slot.buf.numpy()
        Hm = np.zeros((m + 1, m))
        for k in range(m):
            kk = k + 1
            blk = co[k * stride:k * stride + 2 * kk + 1]
            Hm[:kk, k] = blk[:kk] + blk[kk:2 * kk]
            Hm[kk, k] = np.sqrt(max(blk[2 * kk], 0.0))
        scale = max(np.abs(Hm[:m, :m]).max(), 1e-300)
        me = m
        for k in range(m):
            if Hm[k + 1, k] <= 1e-13 * scale:
                me = k + 1
                break
        Hk = Hm[:me, :me]
        ev, S = np.linalg.eigh((Hk + Hk.T) / 2)
        sv = S[:, 0] * (1.0 if S[0, 0] >= 0 else -1.0)
        out = buf.buf.numpy().reshape(-1)
        out[:m] = 0.0
        out[:me] = sv
        out[32:35] = [ev[0], abs(Hm[me, me - 1] * sv[-1]), me]

    def lincomb_dev(self, xs, coef, out=None):
        return self.lincomb(xs, coef.buf.numpy().reshape(-1)[:len(xs)], out=out)

    def lincomb(self, xs, coefs, out=None):
        y = self.empty(xs[0].shape) if out is None else out
        v = sum(float(c) * self._v(x) for c, x in zip(coefs, xs))
        y.buf[: y.size] = torch.from_numpy(np.ascontiguousarray(v))
        return y

    def prof_enable(self, on=True):
        pass

    def prof_summary(self):
        return []


class CpuComplexBackend(CpuBackend):
    """CpuBackend + the MPSK_C128 entry points the interleaved complex host (mpskit_jl_amd.native_cplx) uses: interleaved
    complex DTensors (shape (2 n0, n1, ...), Julia Array{ComplexF64} memory), complex MPO slices, prepared operator, complex
    gauge steps / products / split -- all by NumPy complex arithmetic and the oracle.  Test infrastructure only."""

    _cplx_mode = False

    # ---- memory: interleaved complex
    def upload_c(self, a):
        a = np.asarray(a, dtype=np.complex128)
        flat = np.ravel(a, order="F").view(np.float64).copy()
        return DTensor(torch.from_numpy(flat), (2 * a.shape[0],) + tuple(a.shape[1:]))

    def download_c(self, t):
        flat = t.buf[: t.size].numpy().view(np.complex128)
        return flat.reshape((t.shape[0] // 2,) + tuple(t.shape[1:]), order="F").copy()

    def _set_c(self, t, arr):
        t.buf[: t.size] = torch.from_numpy(np.ravel(np.asarray(arr, dtype=np.complex128), order="F").view(np.float64).copy())
        return t

    def upload_env_c(self, blocks):
        slabs = [np.asarray(b, dtype=np.complex128)[:, k, :] for b in blocks for k in range(np.asarray(b).shape[1])]
        flat = np.concatenate([np.ravel(s, order="F") for s in slabs]).view(np.float64).copy()
        return DTensor(torch.from_numpy(flat), (len(slabs), 2 * slabs[0].shape[0], slabs[0].shape[1]))

    def _env_c(self, t, chis):
        W, Db2, Dk = t.shape
        Db = Db2 // 2
        flat = t.buf[: t.size].numpy().view(np.complex128)
        slabs = [flat[w * Db * Dk:(w + 1) * Db * Dk].reshape((Db, Dk), order="F") for w in range(W)]
        out, o = [], 0
        for chi in chis:
            out.append(np.stack(slabs[o:o + chi], axis=1))
            o += chi
        return out

    def _put_env_c(self, blocks, out=None):
        t = self.upload_env_c(blocks)
        if out is not None:
            out.buf[: t.size] = t.buf[: t.size]
            return out
        return t

    def _set_dtype(self, cplx):
        self._cplx_mode = bool(cplx)

    def mposlice(self, odim, d, chil, chir, blocks, cplx=False):
        s = HostSlice(odim, d, chil, chir, blocks)
        s.cplx = bool(cplx)
        return s

    # ---- operators
    class _Hac:
        def __init__(self, be, H, GL, GR):
            self.be, self.H, self.GL, self.GR = be, H, GL, GR

        def apply(self, x, out=None, nblk=1):
            be, H = self.be, self.H
            be._count("hac_apply_c")
            y = mo.dAC(be.download_c(x), H.oracle, be._env_c(self.GL, H.chil), be._env_c(self.GR, H.chir))
            o = be.empty(self.GL.shape[1], x.shape[1], x.shape[2]) if out is None else out
            return be._set_c(o, y)

    def hac_create(self, H, GL, GR):
        assert getattr(H, "cplx", False), "the stand-in only prepares complex operators"
        return CpuComplexBackend._Hac(self, H, GL, GR)

    def dC(self, GL, GR, c, out=None, cplx=False):
        if not cplx:
            return super().dC(GL, GR, c, out)
        W = GL.shape[0]
        y = mo.dC(self.download_c(c), self._env_c(GL, [1] * W), self._env_c(GR, [1] * W))
        o = self.empty(GL.shape[1], c.shape[1]) if out is None else out
        return self._set_c(o, y)

    def dAC2(self, H1, H2, GL, GR, x2, out=None):
        if not getattr(H1, "cplx", False):
            return super().dAC2(H1, H2, GL, GR, x2, out)
        y = mo.dAC2(self.download_c(x2), H1.oracle, H2.oracle, self._env_c(GL, H1.chil), self._env_c(GR, H2.chir))
        o = self.empty(GL.shape[1], x2.shape[1], GR.shape[2], x2.shape[3]) if out is None else out
        return self._set_c(o, y)

    def transfer_left(self, H, GLin, A, Ab, out=None, cplx=False):
        if not (cplx or getattr(H, "cplx", False)):
            return super().transfer_left(H, GLin, A, Ab, out)
        res = mo.transfer_left(self._env_c(GLin, H.chil), H.oracle, self.download_c(A), self.download_c(Ab))
        return self._put_env_c(res, out)

    def transfer_right(self, H, GRin, A, Ab, out=None, cplx=False):
        if not (cplx or getattr(H, "cplx", False)):
            return super().transfer_right(H, GRin, A, Ab, out)
        res = mo.transfer_right(self._env_c(GRin, H.chir), H.oracle, self.download_c(A), self.download_c(Ab))
        return self._put_env_c(res, out)

    # ---- complex gauge steps / products / split (mpsk_qrpos, mpsk_lqpos, mpsk_gemm, mpsk_tsplit under MPSK_C128)
    def qrpos_c(self, A):
        self._count("qrpos_c")
        q, r = mo.qrpos(self.download_c(A))
        return self.upload_c(q), self.upload_c(r)

    def lqpos_c(self, A):
        self._count("lqpos_c")
        l, q = mo.lqpos(self.download_c(A))
        return self.upload_c(l), self.upload_c(q)

    def gemm_c(self, A, B, transA=False, transB=False, alpha=1.0, beta=0.0, out=None):
        a, b = self.download_c(A), self.download_c(B)
        r = alpha * (a.conj().T if transA else a) @ (b.conj().T if transB else b)
        if out is None:
            return self.upload_c(r)
        if beta != 0.0:
            r = r + beta * self.download_c(out)
        return self._set_c(out, r)

    def gemm_raw(self, tA, tB, M, N, K, alpha, a_ptr, lda, b_ptr, ldb, beta, c_ptr, ldc):
        if not self._cplx_mode:
            return super().gemm_raw(tA, tB, M, N, K, alpha, a_ptr, lda, b_ptr, ldb, beta, c_ptr, ldc)
        ar, ac = (K, M) if tA else (M, K)
        br, bc = (N, K) if tB else (K, N)
        cv = lambda p, ld, cols: _view(p, 2 * ld * cols).view(np.complex128).reshape((ld, cols), order="F")
        a, b, c = cv(a_ptr, lda, ac)[:ar], cv(b_ptr, ldb, bc)[:br], cv(c_ptr, ldc, N)
        r = alpha * (a.conj().T if tA else a) @ (b.conj().T if tB else b)
        c[:M] = r + (beta * c[:M] if beta != 0.0 else 0.0)

    def tsplit_c(self, theta, max_keep=0):
        self._count("tsplit_c")
        a = self.download_c(theta)
        U, S, Vh = np.linalg.svd(a, full_matrices=False)
        k = len(S) if not max_keep else min(len(S), int(max_keep))
        return (self.upload_c(U[:, :k]), self.upload_c(np.diag(S[:k]).astype(complex)), self.upload_c(Vh[:k, :]), S[:k].copy(),
                float(np.sqrt(np.sum(S[k:] ** 2))))
