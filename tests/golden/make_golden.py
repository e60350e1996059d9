#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the REAL reference.

Runs `oracle/_ref/ref_harness` (the reference's own headers compiled where they lie under
/root/reference/src by `make -C oracle _ref`; g++ on purpose) and packs
what it dumps into compressed .npz files.  Only DATA is committed: inputs and the reference's
outputs.  The data file `data/sample_matrix/4x4parsed.txt` (the reference's only shipped
input) is committed gzip-compressed next to them, because the GPU box
has no /root/reference.

Usage (in the build container, where /root/reference exists):
    make -C oracle _ref && python tests/golden/make_golden.py
"""
import gzip
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
HARNESS = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
REF = os.environ.get("MGCR_REFERENCE_ROOT", "/root/reference")


def run(out, *args):
    log = subprocess.run([HARNESS, out, *args], check=True, capture_output=True, text=True)
    return log.stdout, log.stderr


def c(out, name):
    return np.fromfile(os.path.join(out, name + ".bin"), dtype=np.complex128)


def d(out, name, dtype=np.float64):
    return np.fromfile(os.path.join(out, name + ".bin"), dtype=dtype)


def legacy(out):
    """tests/golden/legacy_dense.npz (harness case `legacy`); `python make_golden.py legacy` regenerates it alone"""
    stdout, _ = run(out, "legacy")
    printed, cur = {}, None
    for line in stdout.splitlines():
        if line.startswith("LEGACY "):
            cur = line.split()[1]
            printed[cur] = []
        elif line.startswith("Step ") and cur:
            printed[cur].append(float(line.split("=")[1]))
        elif line.startswith("GCR did not converge") and cur:
            printed[cur + "_final_sq"] = [float(line.split("=")[1])]
    np.savez_compressed(
        os.path.join(HERE, "legacy_dense.npz"), A=c(out, "g16_A"), rhs=c(out, "g16_rhs"), x0=c(out, "g16_x0"),
        x_trunc3=c(out, "g16_x_trunc3"), x_trunc8=c(out, "g16_x_trunc8"), x_zero=c(out, "g16_x_zero"), x_rhs0=c(out, "g16_x_rhs0"),
        printed_trunc3=np.array(printed["trunc3"]), printed_trunc8=np.array(printed["trunc8_tol"]),
        printed_zero=np.array(printed["zero_steps"]), printed_rhs0=np.array(printed["rhs0"]),
        final_sq_trunc3=np.array(printed.get("trunc3_final_sq", [])),
        u_add=c(out, "g16_u_add"), u_amult=c(out, "g16_u_amult"), u_scalars=c(out, "g16_u_scalars"),
        u_normalised=c(out, "g16_u_normalised"), u_matvec=c(out, "g16_u_matvec"))


def main():
    if not os.path.exists(HARNESS):
        sys.exit("build the harness first: make -C oracle _ref")
    out = tempfile.mkdtemp(prefix="mgcr_gold_")
    if sys.argv[1:] == ["legacy"]:
        try:
            legacy(out)
        finally:
            shutil.rmtree(out, ignore_errors=True)
        return
    try:
        stdout, _ = run(out, "sample")
        # the printed history of the first (restart-5) solve, as a cross-check of the spy
        printed = []
        for line in stdout.splitlines():
            if line.startswith("Step "):
                printed.append(float(line.split("=")[1]))
            if line.startswith("GCR converged"):
                break
        sample = dict(
            g1_x=c(out, "g1_x"), g1_Dx=c(out, "g1_Dx"), g1_dirac_x=c(out, "g1_dirac_x"),
            g2_a=c(out, "g2_a"), g2_b=c(out, "g2_b"), g2_scalars=c(out, "g2_scalars"),
            g2_a_plus_alpha_b=c(out, "g2_a_plus_alpha_b"),
            g2_a_minus_alpha_b=c(out, "g2_a_minus_alpha_b"),
            gcr_rhs=c(out, "gcr_rhs"),
            g3_printed=np.array(printed),
        )
        for tag in ["g3_restart5", "g4_restart2", "g5_trunc8", "g6_full", "g10_maxiter0",
                    "g10_x0rand", "g11_right_neumann", "g11_left_neumann", "g3b_complexk"]:
            sample[tag + "_hist"] = d(out, tag + "_hist")
            sample[tag + "_x"] = c(out, tag + "_x")
        np.savez_compressed(os.path.join(HERE, "sample_4x4.npz"), **sample)

        run(out, "hsparse")
        meta = d(out, "g8_meta", np.int32)
        np.savez_compressed(
            os.path.join(HERE, "hsparse.npz"), nb=meta[0], bs=meta[1], nt=meta[2],
            blocks=c(out, "g8_blocks"), rows=d(out, "g8_rows", np.int32),
            cols=d(out, "g8_cols", np.int32), x=c(out, "g8_x"), y=c(out, "g8_y"),
            dense=c(out, "g8_dense"))

        run(out, "mg")
        meta = d(out, "g9_meta", np.int64)
        nblocks, bsz, ne, sub = (int(v) for v in meta[:4])
        N = 3072
        P = c(out, "g9_P").reshape(nblocks, ne, N)
        np.savez_compressed(
            os.path.join(HERE, "mg_4x4.npz"), nblocks=nblocks, block_size=bsz, ne=ne, sub=sub,
            k=0.1, block_map=d(out, "g9_block_map", np.int64).reshape(nblocks, bsz), P=P,
            eigvec0=c(out, "g9_eigvec0"), eigvec1=c(out, "g9_eigvec1"),
            gamma5_eigvec0=c(out, "g9_gamma5_eigvec0"),
            v=c(out, "g9_v"), Rv=c(out, "g9_Rv"), PRv=c(out, "g9_PRv"), AcRv=c(out, "g9_AcRv"),
            Ac_dense=c(out, "g9_Ac_dense").reshape(nblocks * ne, nblocks * ne),
            identities=d(out, "g9_identities"))

        pois = {}
        run(out, "poisson", "32", "10", "p32")
        pois["p32_hist"] = d(out, "p32_hist")
        pois["p32_x"] = c(out, "p32_x")
        run(out, "poisson", "8", "300", "p8t4", "4", "0", "1e-10")
        pois["p8_trunc4_hist"] = d(out, "p8t4_hist")
        pois["p8_trunc4_x"] = c(out, "p8t4_x")
        pois["p8_rhs"] = c(out, "p8t4_rhs")
        # full GCR looses orthogonality and blows up after ~35 steps (no breakdown guard,
        # SURVEY.md §5): pin the well-conditioned prefix only
        run(out, "poisson", "8", "25", "p8full", "0", "0", "1e-10")
        pois["p8_full_hist"] = d(out, "p8full_hist")
        run(out, "poisson", "16", "300", "p16r3", "0", "3", "1e-12")
        pois["p16_restart3_hist"] = d(out, "p16r3_hist")
        pois["p16_restart3_x"] = c(out, "p16r3_x")
        if os.environ.get("MGCR_GOLD_128", "1") == "1":
            run(out, "poisson", "128", "10", "p128")
            pois["p128_hist"] = d(out, "p128_hist")
        np.savez_compressed(os.path.join(HERE, "poisson.npz"), **pois)

        # round 2: input builders, Arnoldi, operator algebra (oracle/ref_harness.cpp G12-G15)
        # parse_data writes ../../data/sample_matrix/parsed.txt relative to the cwd: give it a scratch tree
        scratch = tempfile.mkdtemp(prefix="mgcr_parse_")
        os.makedirs(os.path.join(scratch, "data", "sample_matrix"))
        cwd = os.path.join(scratch, "a", "b")
        os.makedirs(cwd)
        rng = np.random.default_rng(2024)
        n = 12
        lines = ["%%MatrixMarket matrix coordinate complex general", "% 12 x 12 test matrix for parse_data (src/Parse.cpp:9-61)"]
        ents = []
        for r in range(n):
            cs = rng.choice(n, size=int(rng.integers(1, 5)), replace=False)
            for cc in cs:
                mag = 10.0 ** rng.integers(-6, 4)
                ents.append((r + 1, int(cc) + 1, float(rng.uniform(-1, 1) * mag), float(rng.uniform(-1, 1) * mag)))
            if r % 4 == 2:   # a duplicated pair: summed by the triplet constructor
                ents.append((r + 1, int(cs[0]) + 1, 0.5, -0.25))
        ents[0] = (ents[0][0], ents[0][1], 2.0, 0.0)          # an integer-valued entry
        order = rng.permutation(len(ents))
        lines.append("%d %d %d" % (n, n, len(ents)))
        for i in order:
            lines.append("%d %d %.17g %.17g" % ents[i])
        mtx_text = "\n".join(lines) + "\n"
        mtx = os.path.join(scratch, "in.mtx")
        open(mtx, "w").write(mtx_text)
        subprocess.run([HARNESS, out, "builders", mtx], check=True, capture_output=True, text=True, cwd=cwd)
        parsed_text = open(os.path.join(scratch, "data", "sample_matrix", "parsed.txt")).read()
        shutil.rmtree(scratch, ignore_errors=True)
        L = lambda name: d(out, name, np.int64)  # noqa: E731
        np.savez_compressed(
            os.path.join(HERE, "builders.npz"),
            trip_rows=L("g12_trip_rows"), trip_cols=L("g12_trip_cols"), trip_vals=c(out, "g12_trip_vals"),
            meta=L("g12_meta"), ROW=L("g12_ROW"), COL=L("g12_COL"), VAL=c(out, "g12_VAL"),
            dagger_meta=L("g15_dagger_meta"), dagger_ROW=L("g15_dagger_ROW"), dagger_COL=L("g15_dagger_COL"),
            dagger_VAL=c(out, "g15_dagger_VAL"), scalar=c(out, "g15_scalar"), scaled_VAL=c(out, "g15_scaled_VAL"),
            dense_A=c(out, "g15_dense_A"), dense_B=c(out, "g15_dense_B"), dense_AB=c(out, "g15_dense_AB"),
            dense_Adag=c(out, "g15_dense_Adag"), dense_sum_row0=c(out, "g15_dense_sum_row0"),
            mtx_text=np.frombuffer(mtx_text.encode(), np.uint8), parsed_text=np.frombuffer(parsed_text.encode(), np.uint8))
        run(out, "arnoldi")
        np.savez_compressed(os.path.join(HERE, "arnoldi_4x4.npz"), k=0.1, start=c(out, "g14_start"), vec0=c(out, "g14_vec0"),
                            vec1_x0zero=c(out, "g14_vec1_x0zero"))

        legacy(out)

        src = os.path.join(REF, "data", "sample_matrix", "4x4parsed.txt")
        with open(src, "rb") as fi, gzip.GzipFile(
                os.path.join(HERE, "4x4parsed.txt.gz"), "wb", mtime=0) as fo:
            shutil.copyfileobj(fi, fo)
    finally:
        shutil.rmtree(out, ignore_errors=True)
    for f in sorted(os.listdir(HERE)):
        print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
