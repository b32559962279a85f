#!/usr/bin/env python3
"""tools/make_golden_lambda.py -- tests/golden/{se2,se3,ba,ba_robust}_lambda.npz: Lambda and eta as THE REFERENCE'S OWN ASSEMBLY
produces them, together with the per-edge Jacobians / information / errors they were assembled from.

Runs in the development container only: needs oracle/_ref/lambda_dump (make -C oracle lambda_dump), a driver that links
the reference's nonlinear solver against a recording linear solver (oracle/lambda_dump.cpp). The fixtures are numeric
data: inputs (v0, v1, J0, J1, Om, r per edge, vertex dimensions) and expected outputs (the upper block triangle of
Lambda in block-CSC form, eta; for BA also the Levenberg-Marquardt-damped Lambda; ba_robust: the same BA problem with
robust (Huber) edges -- the per-edge weights `w` the reference's kernel returned and the Lambda / eta its robust assembly
branch, BaseTypes_Binary.h:768-848, produced). No reference source text is stored.
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")
EXE = os.path.join(ROOT, "oracle", "_ref", "lambda_dump")


def parse(path):
    edges, records, cur, weights = [], {}, None, []
    with open(path) as f:
        for ln in f:
            t = ln.split()
            if t[0] == "GRAPH":
                kind, nv, ne, rd, d0, d1 = t[1], int(t[2]), int(t[3]), int(t[4]), int(t[5]), int(t[6])
            elif t[0] == "E":
                edges.append(np.array(t[1:], dtype=np.float64))
            elif t[0] == "W":  # robust weight of the edge in the line above (ba_robust)
                weights.append(float(t[1]))
            elif t[0] in ("LAMBDA", "LAMBDA_LM", "LAMBDA_ROBUST"):
                cur = records.setdefault(t[0], dict(blocks=[], nb=int(t[1]), nnzb=int(t[2]), n=int(t[3])))
            elif t[0] == "DIM":
                cur["dim"] = np.array(t[1:], dtype=np.int32)
            elif t[0] == "B":
                cur["blocks"].append((int(t[1]), int(t[2]), np.array(t[3:], dtype=np.float64)))
            elif t[0] == "ETA":
                cur["eta"] = np.array(t[1:], dtype=np.float64)
    e = np.array(edges)
    assert e.shape == (ne, 2 + rd * d0 + rd * d1 + rd * rd + rd), e.shape
    o = 2
    out = dict(kind=kind, nv=nv, rd=rd, d0=d0, d1=d1, v0=e[:, 0].astype(np.int64), v1=e[:, 1].astype(np.int64))
    out["J0"] = e[:, o:o + rd * d0].copy(); o += rd * d0      # column-major rd x d0
    out["J1"] = e[:, o:o + rd * d1].copy(); o += rd * d1
    out["Om"] = e[:, o:o + rd * rd].copy(); o += rd * rd
    out["r"] = e[:, o:o + rd].copy()
    if weights:
        assert len(weights) == ne
        out["w"] = np.array(weights)
    for name, rec in records.items():
        # the driver walks the columns in order and each column's blocks in the matrix' own (ascending row) order
        cols = np.array([b[1] for b in rec["blocks"]])
        rows = np.array([b[0] for b in rec["blocks"]])
        assert np.all(np.diff(cols) >= 0) and rec["nnzb"] == len(rec["blocks"])
        col_ptr = np.zeros(rec["nb"] + 1, dtype=np.int64)
        np.add.at(col_ptr, cols + 1, 1)
        np.cumsum(col_ptr, out=col_ptr)
        sizes = np.array([b[2].size for b in rec["blocks"]], dtype=np.int64)
        blk_off = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)
        sfx = {"LAMBDA": "", "LAMBDA_LM": "_lm", "LAMBDA_ROBUST": ""}[name]
        out.update({"dim": rec["dim"], "col_ptr": col_ptr, "row_idx": rows.astype(np.int64), "blk_off": blk_off,
                    "vals" + sfx: np.concatenate([b[2] for b in rec["blocks"]]), "eta" + sfx: rec["eta"]})
    return out


def main():
    assert os.path.exists(EXE), "build it first: make -C oracle lambda_dump"
    os.makedirs(OUT, exist_ok=True)
    with tempfile.TemporaryDirectory() as tmp:
        for kind in ("se2", "se3", "ba", "ba_robust"):
            txt = os.path.join(tmp, kind + ".txt")
            subprocess.run([EXE, kind, txt], check=True, cwd=tmp, capture_output=True)
            d = parse(txt)
            np.savez_compressed(os.path.join(OUT, kind + "_lambda.npz"), **d)
            print("%s: %d vertices, %d edges, %d upper blocks, n = %d -> %s_lambda.npz" % (
                kind, d["nv"], d["v0"].size, d["row_idx"].size, d["eta"].size, kind))


if __name__ == "__main__":
    main()
