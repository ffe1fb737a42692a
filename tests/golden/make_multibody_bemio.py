#!/usr/bin/env python3
"""Writes a synthetic THREE-body BEMIO file (run in the BUILD container only; needs gcc + libhdf5 from /opt/conda).

The reference snapshot holds no multi-body BEMIO blob (rm3.h5 / oswec.h5 / f3of.h5 are missing), so the multi-body
dataset shapes H5FileInfo::ReadH5Data reads (src/h5fileinfo.cpp:41-90: body1..N with added_mass/inf_freq {6,6N},
impulse_response_fun/K {6,6N,S}, excitation {6,1,nw} / {6,1,n}) and the string-valued water depth "infinite"
(:207-220) are exercised with a generated file of exactly those shapes:

  tests/golden/three_body.h5        water_depth = fixed-length string "infinite"
  tests/golden/three_body_vlen.h5   the same file with a variable-length string
  tests/golden/three_body_bemio.npz the same arrays, flat (what the raw-array setters get in the test)

Arrays come from hydrochrono_amd.synthetic.many_body_case(3, ...) (deterministic); the HDF5 files are written by a small
C program built here (HDF5 C API -- there is no h5py in the image)."""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from hydrochrono_amd.synthetic import many_body_case  # noqa: E402

WRITER = r'''
#include <hdf5.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
/* input: a list of records  name\n rank d0 d1 ..\n <raw little-endian doubles>  ; strings: name\n -1 len\n bytes */
static void mkgroups(hid_t f, const char* path) {
    char buf[512];
    strncpy(buf, path, sizeof buf - 1);
    buf[sizeof buf - 1] = 0;
    for (char* p = buf + 1; *p; ++p)
        if (*p == '/') {
            *p = 0;
            if (H5Lexists(f, buf, H5P_DEFAULT) <= 0) H5Gclose(H5Gcreate2(f, buf, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT));
            *p = '/';
        }
}
int main(int argc, char** argv) {
    if (argc < 4) return 2;
    const int vlen = atoi(argv[3]);
    FILE* in = fopen(argv[1], "rb");
    hid_t f = H5Fcreate(argv[2], H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT);
    char name[512];
    while (fgets(name, sizeof name, in)) {
        name[strcspn(name, "\n")] = 0;
        int rank;
        if (fscanf(in, "%d", &rank) != 1) return 3;
        mkgroups(f, name);
        if (rank < 0) {
            int len;
            if (fscanf(in, "%d", &len) != 1) return 3;
            fgetc(in);
            char* s = calloc(len + 1, 1);
            if (fread(s, 1, len, in) != (size_t)len) return 3;
            hid_t ty = H5Tcopy(H5T_C_S1), sp = H5Screate(H5S_SCALAR);
            if (vlen) H5Tset_size(ty, H5T_VARIABLE); else H5Tset_size(ty, len);
            hid_t ds = H5Dcreate2(f, name, ty, sp, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
            if (vlen) { const char* p = s; H5Dwrite(ds, ty, H5S_ALL, H5S_ALL, H5P_DEFAULT, &p); }
            else H5Dwrite(ds, ty, H5S_ALL, H5S_ALL, H5P_DEFAULT, s);
            H5Dclose(ds); H5Sclose(sp); H5Tclose(ty); free(s);
        } else {
            hsize_t dims[8]; size_t n = 1;
            for (int i = 0; i < rank; ++i) { long long d; if (fscanf(in, "%lld", &d) != 1) return 3; dims[i] = d; n *= d; }
            fgetc(in);
            double* buf = malloc(n * sizeof(double));
            if (fread(buf, sizeof(double), n, in) != n) return 3;
            hid_t sp = rank ? H5Screate_simple(rank, dims, NULL) : H5Screate(H5S_SCALAR);
            hid_t ds = H5Dcreate2(f, name, H5T_NATIVE_DOUBLE, sp, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
            H5Dwrite(ds, H5T_NATIVE_DOUBLE, H5S_ALL, H5S_ALL, H5P_DEFAULT, buf);
            H5Dclose(ds); H5Sclose(sp); free(buf);
        }
    }
    H5Fclose(f); fclose(in);
    return 0;
}
'''


def bemio_records(case):
    """The datasets of a BEMIO file for `case` (hydrochrono_amd.synthetic.many_body_case layout), in file order:
    [(name, array | str)] and the same arrays flat by name."""
    N = case["N"]
    recs, flat = [], {}

    def num(name, arr, shape):
        a = np.ascontiguousarray(np.asarray(arr, dtype="<f8")).reshape(shape)
        recs.append((name, a))
        flat[name.strip("/")] = a

    num("/simulation_parameters/rho", [case["rho"]], ())
    num("/simulation_parameters/g", [case["g"]], ())
    recs.append(("/simulation_parameters/water_depth", "infinite"))
    w = case["bodies"][0]["w"]
    nw = len(w)
    num("/simulation_parameters/w", w, (nw, 1))
    for b, bd in enumerate(case["bodies"]):
        p = f"/body{b + 1}"
        S, n_exc = len(bd["rirf_t"]), len(bd["ex_irf_t"])
        num(p + "/properties/disp_vol", [bd["disp_vol"]], ())
        num(p + "/properties/body_number", [b + 1.0], ())
        num(p + "/properties/cg", bd["cg"], (3,))
        num(p + "/properties/cb", bd["cb"], (3,))
        num(p + "/hydro_coeffs/linear_restoring_stiffness", bd["lin"], (6, 6))
        num(p + "/hydro_coeffs/added_mass/inf_freq", bd["added_mass_inf"], (6, 6 * N))
        num(p + "/hydro_coeffs/radiation_damping/impulse_response_fun/t", bd["rirf_t"], (S,))
        num(p + "/hydro_coeffs/radiation_damping/impulse_response_fun/K", bd["rirf_K"], (6, 6 * N, S))
        num(p + "/hydro_coeffs/excitation/mag", bd["ex_mag"], (6, 1, nw))
        num(p + "/hydro_coeffs/excitation/phase", bd["ex_phase"], (6, 1, nw))
        num(p + "/hydro_coeffs/excitation/impulse_response_fun/t", bd["ex_irf_t"], (n_exc,))
        num(p + "/hydro_coeffs/excitation/impulse_response_fun/f", bd["ex_irf_f"], (6, 1, n_exc))
    return recs, flat


def write_bemio(case, paths_vlen):
    """Writes `case` as BEMIO HDF5 files: paths_vlen = [(path, variable-length water-depth string?)].  Needs gcc and libhdf5 (the small C
    writer above is built in a temporary directory; /opt/conda holds HDF5 in this image).  Any size: bench.py's `init` block writes the
    C3-size file (64 bodies x 1024 IRF samples, 1.2 GB) to /tmp with it.  Returns the flat arrays."""
    recs, flat = bemio_records(case)
    with tempfile.TemporaryDirectory() as tmp:
        spec, src, exe = (os.path.join(tmp, x) for x in ("spec.bin", "w.c", "w"))
        with open(spec, "wb") as f:
            for name, a in recs:
                if isinstance(a, str):
                    f.write(f"{name}\n-1 {len(a)}\n".encode() + a.encode())
                else:
                    f.write((name + "\n" + " ".join([str(a.ndim)] + [str(d) for d in a.shape]) + "\n").encode())
                    f.write(memoryview(a).cast("B") if a.ndim else a.tobytes())
        open(src, "w").write(WRITER)
        subprocess.run(["gcc", "-O1", src, "-o", exe, "-I/opt/conda/include", "-L/opt/conda/lib", "-lhdf5", "-Wl,-rpath,/opt/conda/lib"], check=True)
        for path, vlen in paths_vlen:
            subprocess.run([exe, spec, path, "1" if vlen else "0"], check=True)
    return flat


def write_case(N, seed, stem, vlen_copy):
    S, nw, n_exc = 32, 16, 33
    case = many_body_case(N, S=S, dt_rirf=0.02, n_exc=n_exc, dt_exc=0.05, nw=nw, seed=seed)
    paths = [(os.path.join(HERE, stem + ".h5"), False)] + ([(os.path.join(HERE, stem + "_vlen.h5"), True)] if vlen_copy else [])
    flat = write_bemio(case, paths)
    np.savez_compressed(os.path.join(HERE, stem + "_bemio.npz"), **flat)
    print(f"wrote {stem}.h5, {stem}_bemio.npz" + (f", {stem}_vlen.h5" if vlen_copy else ""))


def main():
    write_case(3, 31337, "three_body", True)
    # four bodies (D = 24: a multiple of 8, the scalar-tracker form of the look-ahead pass): the file the multi-shard C++ test
    # (tests/cpp/shards_test.cpp) reads with 1, 2 and 4 row shards
    write_case(4, 4242, "four_body", False)


if __name__ == "__main__":
    main()
