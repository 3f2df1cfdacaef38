#!/usr/bin/env python3
"""Regenerate tests/golden/*.npz from the REAL reference (authoring container only).

Runs oracle/_ref/ref_driver (built by `make -C oracle/ref` from the reference
sources where they lie under /root/reference) on the reference's five example
matrices and packs its raw outputs into one .npz per matrix.  The .mtx inputs
are public SuiteSparse data files held by the reference's example/ directory
and are copied next to the vectors as input fixtures.  Only data is committed;
no reference source text is.

Keys of each <name>.npz (all produced by reference code, see ref_driver.cpp):
  dims                  [rows, cols, nnz_in_header]
  f32_row_ptr/col_idx/val   CSR view of SparseMatrix<float>::ellpack_encode()
  i32_row_ptr/col_idx/val   same for SparseMatrix<int> (BFS element type)
  gold_x1               Gold<float>::spmv, x=1, y=0, alpha=1, beta=0  (app/spmv.cpp:117-120,144)
  gold_xmod             Gold<float>::spmv, x[i]=1+i%7
  gold_ab               Gold<float>::spmv, x[i]=1+i%7, y=3, alpha=2, beta=.5 (quirk A-4)
  ell_hw                [cl_height, cl_width] of cl_encode(..., no padding)
  kern_spmv_x1          Lift glb-sdp spmv kernel, x=1,y=0,alpha=1,beta=0
  kern_spmv_ab          Lift glb-sdp spmv kernel, x[i]=1+i%7, y[i]=i%5, alpha=2, beta=.5
  sssp_meta             [kernel launches incl. confirming one, converged?]
  sssp_first/final      vector after launch 1 / at termination
  bfs_meta, bfs_first/final   same for the (or,and) kernel
  pr_row_ptr/col_idx/val      SparseMatrix<float> rows after pagerank_normalise(0.85, 0) (+ int narrowing)
  pr_meta, pr_first/final     PageRank app loop on the Lift pr kernel (app/pr.cpp constants)
  scc_row_ptr/col_idx/val     SparseMatrix<int> rows after scc_normalise()
  scc_meta, scc_first/final   SCC app loop on the Lift (max,min) kernel (app/scc.cpp constants)
"""
import glob
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("SH_REFERENCE", "/root/reference")
DRIVER = os.path.join(ROOT, "oracle", "_ref", "ref_driver")
DT = {"i32": np.int32, "f32": np.float32}


def main():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle", "ref")])
    for name in ["matrix", "matrix2", "matrix3", "matrix4", "matrix5"]:
        src = os.path.join(REF, "example", name + ".mtx")
        with tempfile.TemporaryDirectory() as tmp:
            prefix = os.path.join(tmp, name)
            subprocess.check_call([DRIVER, src, prefix], stdout=subprocess.DEVNULL,
                                  stderr=subprocess.DEVNULL)
            arrays = {}
            for path in sorted(glob.glob(prefix + ".*")):
                _, key, dt = os.path.basename(path).split(".")
                arrays[key] = np.fromfile(path, dtype=DT[dt])
            np.savez_compressed(os.path.join(HERE, name + ".npz"), **arrays)
        shutil.copyfile(src, os.path.join(HERE, name + ".mtx"))
        os.chmod(os.path.join(HERE, name + ".mtx"), 0o644)
        print(name, {k: v.shape for k, v in arrays.items() if k.startswith(("gold_x1", "sssp_meta", "bfs_meta"))},
              "sum(gold_x1)=", float(arrays["gold_x1"].astype(np.float64).sum()),
              "sssp", arrays["sssp_meta"].tolist(), "bfs", arrays["bfs_meta"].tolist())


if __name__ == "__main__":
    sys.exit(main())
