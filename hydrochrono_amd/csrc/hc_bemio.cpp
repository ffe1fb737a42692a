// hc_bemio.cpp -- BEMIO-HDF5 reader / result-file writer (libhc_bemio.so; optional, needs libhdf5).
//
// Reads exactly the datasets H5FileInfo::ReadH5Data reads (src/h5fileinfo.cpp:35-90) for body1..bodyN, unscaled: the caller
// (hc_setup.cpp: hc_load_bemio_h5, hc_h5_read) hands them to the raw-array setters of the C ABI, which apply the reference's
// rho / rho*g scaling.  HDF5 C API only; dataset element order is the file's row-major order, as the setters expect.
// The library calls nothing of libhydrochrono_amd.so: data in, data out (either flavour of the main library loads it).
#include <hdf5.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <stdexcept>
#include <string>
#include <vector>

#include "hc_h5data.hpp"

namespace { constexpr int HC_OK = 0, HC_ERR_RUNTIME = 1; }  // (status values of include/hydrochrono_amd.h)

namespace {

struct H5Error : std::runtime_error {
    using std::runtime_error::runtime_error;
};

struct File {
    hid_t id;
    explicit File(const char* path) {
        H5Eset_auto2(H5E_DEFAULT, nullptr, nullptr);
        id = H5Fopen(path, H5F_ACC_RDONLY, H5P_DEFAULT);
        if (id < 0) throw H5Error(std::string("Unable to open/read HDF5 hydro data file: ") + path);
    }
    ~File() {
        if (id >= 0) H5Fclose(id);
    }
};

std::vector<double> read_doubles(hid_t file, const std::string& name, std::vector<hsize_t>* dims_out = nullptr) {
    hid_t ds = H5Dopen2(file, name.c_str(), H5P_DEFAULT);
    if (ds < 0) throw H5Error("missing dataset " + name);
    hid_t sp      = H5Dget_space(ds);
    const int rk  = H5Sget_simple_extent_ndims(sp);
    std::vector<hsize_t> dims(rk > 0 ? rk : 0);
    if (rk > 0) H5Sget_simple_extent_dims(sp, dims.data(), nullptr);
    size_t n = 1;
    for (auto d : dims) n *= static_cast<size_t>(d);
    std::vector<double> out(n);
    const herr_t rc = H5Dread(ds, H5T_NATIVE_DOUBLE, H5S_ALL, H5S_ALL, H5P_DEFAULT, out.data());
    H5Sclose(sp);
    H5Dclose(ds);
    if (rc < 0) throw H5Error("cannot read dataset " + name);
    if (dims_out) *dims_out = dims;
    return out;
}

// InitScalar (src/h5fileinfo.cpp:197-226): numeric scalar, or the string "infinite" -> +inf
double read_scalar(hid_t file, const std::string& name) {
    hid_t ds = H5Dopen2(file, name.c_str(), H5P_DEFAULT);
    if (ds < 0) throw H5Error("missing dataset " + name);
    hid_t ty             = H5Dget_type(ds);
    const H5T_class_t cl = H5Tget_class(ty);
    double v             = 0.0;
    if (cl == H5T_STRING) {
        const size_t sz = H5Tget_size(ty);
        std::vector<char> buf(sz + 1, 0);
        if (H5Tis_variable_str(ty) > 0) {
            char* p = nullptr;
            if (H5Dread(ds, ty, H5S_ALL, H5S_ALL, H5P_DEFAULT, &p) >= 0 && p) {
                if (std::strcmp(p, "infinite") == 0) v = std::numeric_limits<double>::infinity();
                H5free_memory(p);
            }
        } else if (H5Dread(ds, ty, H5S_ALL, H5S_ALL, H5P_DEFAULT, buf.data()) >= 0) {
            if (std::string(buf.data()) == "infinite") v = std::numeric_limits<double>::infinity();
        }
    } else {
        if (H5Dread(ds, H5T_NATIVE_DOUBLE, H5S_ALL, H5S_ALL, H5P_DEFAULT, &v) < 0) {
            H5Tclose(ty);
            H5Dclose(ds);
            throw H5Error("cannot read dataset " + name);
        }
    }
    H5Tclose(ty);
    H5Dclose(ds);
    return v;
}

// body<b + 1> of an N-body file; with_K = false leaves the radiation IRF tensor {6, 6N, S} unread (q.K empty; q.rirf_t is read)
void read_body(hid_t file, int N, int b, size_t nw, bool with_K, hc_h5data::Body& q) {
    const int D            = 6 * N;
    const std::string body = "body" + std::to_string(b + 1);
    q.disp_vol       = read_scalar(file, body + "/properties/disp_vol");
    const auto cg    = read_doubles(file, body + "/properties/cg");
    const auto cb    = read_doubles(file, body + "/properties/cb");
    if (cg.size() < 3 || cb.size() < 3) throw H5Error(body + ": cg/cb must have 3 entries");
    const auto lin = read_doubles(file, body + "/hydro_coeffs/linear_restoring_stiffness");
    if (lin.size() != 36) throw H5Error(body + ": linear_restoring_stiffness must be 6x6");
    for (int k = 0; k < 3; ++k) {
        q.cg[k] = cg[static_cast<size_t>(k)];
        q.cb[k] = cb[static_cast<size_t>(k)];
    }
    for (int k = 0; k < 36; ++k) q.lin[k] = lin[static_cast<size_t>(k)];
    q.ainf = read_doubles(file, body + "/hydro_coeffs/added_mass/inf_freq");
    if (q.ainf.size() != static_cast<size_t>(6) * D) throw H5Error(body + ": added_mass/inf_freq must be 6 x 6N");
    q.rirf_t = read_doubles(file, body + "/hydro_coeffs/radiation_damping/impulse_response_fun/t");
    q.K.clear();
    if (with_K) {
        std::vector<hsize_t> kd;
        q.K = read_doubles(file, body + "/hydro_coeffs/radiation_damping/impulse_response_fun/K", &kd);
        if (kd.size() != 3 || kd[0] != 6 || kd[1] != static_cast<hsize_t>(D) || kd[2] != q.rirf_t.size())
            throw H5Error(body + ": impulse_response_fun/K must be {6, 6N, len(t)}");
    }
    q.mag   = read_doubles(file, body + "/hydro_coeffs/excitation/mag");
    q.phase = read_doubles(file, body + "/hydro_coeffs/excitation/phase");
    if (q.mag.size() != 6 * nw || q.phase.size() != 6 * nw) throw H5Error(body + ": excitation mag/phase must be {6,1,len(w)}");
    q.exc_t = read_doubles(file, body + "/hydro_coeffs/excitation/impulse_response_fun/t");
    q.exc_f = read_doubles(file, body + "/hydro_coeffs/excitation/impulse_response_fun/f");
    if (q.exc_f.size() != 6 * q.exc_t.size()) throw H5Error(body + ": excitation impulse_response_fun/f must be {6,1,len(t)}");
}

void read_header(hid_t file, hc_h5data* out) {
    out->rho         = read_scalar(file, "simulation_parameters/rho");
    out->g           = read_scalar(file, "simulation_parameters/g");
    out->water_depth = read_scalar(file, "simulation_parameters/water_depth");
    out->w           = read_doubles(file, "simulation_parameters/w");
}

}  // namespace

// The file's content for body1 .. bodyN, with the shape checks of the reader (src/h5fileinfo.cpp:35-90, 287-294): everything at
// once -- the host-side view of a file (H5FileInfo / HydroData, include/hydroc_amd/h5fileinfo.h).
extern "C" int hc_bemio_read(const char* path, int N, hc_h5data* out, char* err, size_t errlen) {
    try {
        if (N <= 0) throw H5Error("num_bodies must be positive");
        File f(path);
        read_header(f.id, out);
        out->bodies.assign(static_cast<size_t>(N), hc_h5data::Body{});
        for (int b = 0; b < N; ++b) read_body(f.id, N, b, out->w.size(), true, out->bodies[static_cast<size_t>(b)]);
    } catch (const std::exception& e) {
        if (err && errlen) std::snprintf(err, errlen, "%s", e.what());
        return HC_ERR_RUNTIME;
    }
    return HC_OK;
}

// One body at a time, for the ingest into a device context (hc_load_bemio_h5): the caller reads body b, hands it to the setters and
// lets go of it before it asks for the next, so that the host never holds more than one body's {6, 6N, S} tensor -- all bodies at
// once are (6N)^2 S doubles, tens of gigabytes for a few hundred bodies, and every shard context of a row-sharded system reads the
// file.  body < 0: the simulation parameters only (out->rho, g, water_depth, w).  with_K = 0: without the radiation IRF tensor (a
// shard context needs it for the bodies it owns only).
extern "C" int hc_bemio_read_body(const char* path, int N, int body, int with_K, hc_h5data* out, char* err, size_t errlen) {
    try {
        if (N <= 0 || body >= N) throw H5Error("body index out of range");
        File f(path);
        read_header(f.id, out);
        out->bodies.clear();
        if (body >= 0) {
            out->bodies.assign(1, hc_h5data::Body{});
            read_body(f.id, N, body, out->w.size(), with_K != 0, out->bodies[0]);
        }
    } catch (const std::exception& e) {
        if (err && errlen) std::snprintf(err, errlen, "%s", e.what());
        return HC_ERR_RUNTIME;
    }
    return HC_OK;
}


// ---------------------------------------------------------------------------------------------------------------
// Result-file side: SimulationExporter::WriteIrregularInputs (src/simulation_exporter.cpp:365-393) -- the spectrum and the
// free-surface table of the attached irregular wave model under /inputs/simulation/waves/irregular, same dataset and
// attribute names, so the reference's comparison tooling can read the new path's outputs.
// ---------------------------------------------------------------------------------------------------------------
namespace {

hid_t require_group(hid_t parent, const char* name) {
    if (H5Lexists(parent, name, H5P_DEFAULT) > 0) return H5Gopen2(parent, name, H5P_DEFAULT);
    return H5Gcreate2(parent, name, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
}

void write_string_attr(hid_t obj, const char* name, const char* value) {
    hid_t type = H5Tcopy(H5T_C_S1);
    H5Tset_size(type, std::strlen(value) + 1);
    hid_t space = H5Screate(H5S_SCALAR);
    if (H5Aexists(obj, name) > 0) H5Adelete(obj, name);
    hid_t attr = H5Acreate2(obj, name, type, space, H5P_DEFAULT, H5P_DEFAULT);
    if (attr >= 0) {
        H5Awrite(attr, type, value);
        H5Aclose(attr);
    }
    H5Sclose(space);
    H5Tclose(type);
}

void write_vector(hid_t group, const char* name, const std::vector<double>& v) {
    if (v.empty()) return;  // the reference skips empty vectors
    if (H5Lexists(group, name, H5P_DEFAULT) > 0) H5Ldelete(group, name, H5P_DEFAULT);
    hsize_t dims[1] = {static_cast<hsize_t>(v.size())};
    hid_t space     = H5Screate_simple(1, dims, nullptr);
    hid_t ds        = H5Dcreate2(group, name, H5T_NATIVE_DOUBLE, space, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    if (ds < 0) {
        H5Sclose(space);
        throw H5Error(std::string("cannot create dataset ") + name);
    }
    H5Dwrite(ds, H5T_NATIVE_DOUBLE, H5S_ALL, H5S_ALL, H5P_DEFAULT, v.data());
    H5Dclose(ds);
    H5Sclose(space);
}

}  // namespace

extern "C" int hc_bemio_export_irregular(const char* path, const double* f_hz, const double* S_f, int nf, const double* t_eta, const double* eta_t, int nt,
                                         char* err, size_t errlen) {
    try {
        const std::vector<double> f(f_hz, f_hz + (nf > 0 ? nf : 0)), S(S_f, S_f + (nf > 0 ? nf : 0)), t(t_eta, t_eta + (nt > 0 ? nt : 0)),
            eta(eta_t, eta_t + (nt > 0 ? nt : 0));
        H5Eset_auto2(H5E_DEFAULT, nullptr, nullptr);
        hid_t file = H5Fopen(path, H5F_ACC_RDWR, H5P_DEFAULT);
        if (file < 0) file = H5Fcreate(path, H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT);
        if (file < 0) throw H5Error(std::string("cannot open or create ") + path);
        hid_t g1 = require_group(file, "inputs");
        hid_t g2 = require_group(g1, "simulation");
        hid_t g3 = require_group(g2, "waves");
        hid_t g  = require_group(g3, "irregular");
        write_vector(g, "frequencies_hz", f);
        if (nf) write_string_attr(g, "frequencies_hz.units", "Hz");
        write_vector(g, "spectral_densities", S);
        if (nf) {
            write_string_attr(g, "spectral_densities.units", "m^2/Hz");
            write_string_attr(g, "spectral_densities.convention", "JONSWAP (if gamma>1), else PM");
        }
        write_vector(g, "free_surface_time", t);
        if (nt) write_string_attr(g, "free_surface_time.units", "s");
        write_vector(g, "free_surface_eta", eta);
        if (nt) {
            write_string_attr(g, "free_surface_eta.units", "m");
            write_string_attr(g, "free_surface_eta.location", "x=0,y=0,z=0 (assumed)");
        }
        H5Gclose(g);
        H5Gclose(g3);
        H5Gclose(g2);
        H5Gclose(g1);
        H5Fclose(file);
    } catch (const std::exception& e) {
        if (err && errlen) std::snprintf(err, errlen, "%s", e.what());
        return HC_ERR_RUNTIME;
    }
    return HC_OK;
}
