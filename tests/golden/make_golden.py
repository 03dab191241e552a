#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the REAL reference.

Runs only in the build container, where /root/reference is mounted:

    make -C oracle ref            # compiles the reference from its own sources + oracle/ref_driver.cc
    python tests/golden/make_golden.py

Outputs (data only -- inputs and expected outputs; no reference source text):
    basic.json.gz        boundary::enforce table, mixed_dist_product::evaluate_log tables, ladders
    trace{1..6}.json.gz parallel_tempering_chains traces (5, 6: ladder evolving, evolve_temps) (recorded RNG tapes, scripted proposals,
                         expected per-step per-rung state / llike / lpost / history size)
    lisa_init_rows.json  the 31 prior-draw rows of the reference's own golden file
                         test/exampleLISA/exampleLISA_test_0_t0.dat (i, lpost, llike, params)
    gauss_target.json    correlated-Gaussian log-likelihood values following
                         cython/exampleGaussian.py:53-54,61,103-109 evaluated with numpy
"""
import gzip
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
DRIVER = os.path.join(ROOT, "oracle", "_ref", "ptm_ref_driver")
REF = os.environ.get("PTM_REFERENCE", "/root/reference")


def run(*args):
    return subprocess.check_output([DRIVER, *args]).decode()


def dump_gz(name, text):
    json.loads(text)  # validate
    with gzip.GzipFile(os.path.join(HERE, name), "wb", mtime=0) as f:
        f.write(text.encode())


def lisa_rows():
    rows = []
    with open(os.path.join(REF, "test", "exampleLISA", "exampleLISA_test_0_t0.dat")) as f:
        for line in f:
            if line.startswith("#"):
                if rows:
                    break
                continue
            head, tail = line.split(":")
            h = head.split()
            t = [float(v) for v in tail.split()]
            i = int(h[0])
            if i > 0:
                break
            rows.append({"i": i, "lpost": float(h[1]), "llike": float(h[2]), "x": t[:-1], "invtemp": t[-1]})
    return rows


def gauss_target():
    # the reference's correlated Gaussian target (exampleGaussian.py setup()/evaluate_log()) in numpy
    out = []
    rng = np.random.default_rng(20260101)
    for D in (2, 5, 16, 32):
        A = rng.uniform(-1, 1, (D, D))
        cov = A.T @ A / D + 0.1 * np.eye(D)
        lndetcov = np.linalg.slogdet(cov)[1]
        like0 = -0.5 * (D * np.log(2 * np.pi) + lndetcov)
        invcov = np.linalg.inv(cov)
        xs = rng.normal(size=(12, D)) * np.sqrt(np.diag(cov)) * rng.choice([0.3, 1, 4, 30], size=(12, 1))
        ll = [float(like0 - 0.5 * np.dot(p, np.dot(invcov, p))) for p in xs]
        out.append({"D": D, "cov": cov.ravel().tolist(), "invcov": invcov.ravel().tolist(), "like0": float(like0),
                    "x": xs.tolist(), "llike": ll})
    return out


def main():
    if not os.path.exists(DRIVER):
        sys.exit("build the reference first: make -C oracle ref")
    dump_gz("basic.json.gz", run("golden-basic"))
    for i in (1, 2, 3, 4, 5, 6):
        dump_gz("trace%d.json.gz" % i, run("golden-trace", str(i)))
    with open(os.path.join(HERE, "lisa_init_rows.json"), "w") as f:
        json.dump(lisa_rows(), f, indent=0)
    with open(os.path.join(HERE, "gauss_target.json"), "w") as f:
        json.dump(gauss_target(), f)
    for n in sorted(os.listdir(HERE)):
        print("%9d  %s" % (os.path.getsize(os.path.join(HERE, n)), n))


if __name__ == "__main__":
    main()
