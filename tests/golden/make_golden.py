"""Generates the committed golden fixtures (small seeded input/output vectors of the hot path).

The reference (Julia) cannot be executed in this image and its own tests hold no input/output
vectors for the hot-path contractions, so:
  * reference_recorded.json holds the numbers the reference's own docs/tests record (data copied as
    numbers with their file:line provenance) -- these pin the oracle END-TO-END;
  * hotpath_vectors.npz holds seeded inputs and the oracle's outputs for every operator -- a
    regression pin for the oracle and the fixture the GPU parity tests compare against.
Run:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle"))
import mpskit_oracle as mo  # noqa: E402

RECORDED = {
    "dmrg_tfi_obc_L20_D10": {
        "value": -20.40021786703, "digits": 11,
        "what": "DMRG energy, transverse-field Ising |g|=0.5, OBC, L=20, D=10 (H = -sum ZZ - g sum X, Pauli)",
        "source": "docs/src/examples/quantum1d/3.ising-dqpt/index.md:34-48"},
    "vumps_tfi_inf_D10": {
        "value": -1.063544409973, "digits": 12,
        "what": "VUMPS energy density, infinite transverse-field Ising |g|=0.5, D=10",
        "source": "docs/src/examples/quantum1d/3.ising-dqpt/index.md:105-118"},
    "haldane_gap": {
        "value": 0.41047925, "tol": 1e-4,
        "what": "S=1 Heisenberg Haldane gap (known answer asserted by the reference's tests)",
        "source": "test/algorithms.jl:209"},
    "vumps_heisenberg_s1_energy_density": {
        "value": -1.401484038967, "digits": 9,
        "what": "S=1 Heisenberg energy density (SU(2) VUMPS + CG, large D)",
        "source": "docs/src/examples/quantum1d/2.haldane/index.md:430"},
}


def main():
    with open(os.path.join(HERE, "reference_recorded.json"), "w") as f:
        json.dump(RECORDED, f, indent=1, sort_keys=True)

    rng = np.random.default_rng(20240213)
    out = {}
    # case A: Heisenberg slice, chi = 1
    D, d = 12, 2
    H = mo.heisenberg_mpo(0.5)[0]
    GL = [rng.random((D, 1, D)) for _ in range(5)]
    GR = [rng.random((D, 1, D)) for _ in range(5)]
    x = rng.random((D, d, D))
    c = rng.random((D, D))
    x2 = rng.random((D, d, D, d))
    A, Ab = rng.random((D, d, D)), rng.random((D, d, D))
    out.update(A_GL=np.stack(GL), A_GR=np.stack(GR), A_x=x, A_c=c, A_x2=x2, A_A=A, A_Ab=Ab,
               A_dAC=mo.dAC(x, H, GL, GR), A_dC=mo.dC(c, GL, GR), A_dAC2=mo.dAC2(x2, H, H, GL, GR),
               A_tl=np.stack(mo.transfer_left(GL, H, A, Ab)), A_tr=np.stack(mo.transfer_right(GR, H, A, Ab)))
    # case B: two-site-decomposed TFI (chi = [1, r, 1]), ragged bond dims
    Hb = mo.tfi_twosite_mpo(1.3)[0]
    chis = Hb.chil
    Dl, Dr = 5, 9
    GLb = [rng.standard_normal((Dl, ch, Dl)) for ch in chis]
    GRb = [rng.standard_normal((Dr, ch, Dr)) for ch in chis]
    xb = rng.standard_normal((Dl, 2, Dr))
    out.update(B_chis=np.array(chis), B_x=xb, B_dAC=mo.dAC(xb, Hb, GLb, GRb),
               **{f"B_GL{i}": g for i, g in enumerate(GLb)}, **{f"B_GR{i}": g for i, g in enumerate(GRb)},
               **{f"B_O_{i}_{j}": np.asarray(Hb.dense(i, j)) for (i, j) in Hb.keys()})
    # gauge steps
    M = rng.random((24, 10))
    q, r = mo.qrpos(M)
    l, qq = mo.lqpos(M.T.copy())
    th = rng.random((6, 2, 7, 2))
    U, S, Vh, err = mo.tsvd(th, truncdim=5)
    out.update(G_M=M, G_Q=q, G_R=r, G_L=l, G_LQ=qq, G_theta=th, G_S=S, G_err=np.array(err))
    np.savez_compressed(os.path.join(HERE, "hotpath_vectors.npz"), **out)
    print("wrote", len(out), "arrays")


if __name__ == "__main__":
    main()
