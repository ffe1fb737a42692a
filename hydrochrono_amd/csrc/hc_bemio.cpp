// hc_bemio.cpp -- BEMIO-HDF5 ingest (libhc_bemio.so; optional, needs libhdf5).
//
// Reads exactly the datasets H5FileInfo::ReadH5Data reads (src/h5fileinfo.cpp:35-90) for body1..bodyN and hands
// them, unscaled, to the raw-array setters of the C ABI, which apply the reference's rho / rho*g scaling.
// HDF5 C API only; dataset element order is the file's row-major order, as the setters expect.
#include <hdf5.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/hydrochrono_amd.h"
#include "hc_h5data.hpp"

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

void ok(hc_ctx* ctx, int rc) {
    if (rc != HC_OK) throw H5Error(hc_last_error(ctx));
}

}  // namespace

// The file's content for body1 .. bodyN, with the shape checks of the reader (src/h5fileinfo.cpp:35-90, 287-294).
extern "C" int hc_bemio_read(const char* path, int N, hc_h5data* out, char* err, size_t errlen) {
    try {
        if (N <= 0) throw H5Error("num_bodies must be positive");
        File f(path);
        out->rho         = read_scalar(f.id, "simulation_parameters/rho");
        out->g           = read_scalar(f.id, "simulation_parameters/g");
        out->water_depth = read_scalar(f.id, "simulation_parameters/water_depth");
        out->w           = read_doubles(f.id, "simulation_parameters/w");
        const size_t nw  = out->w.size();
        const int D      = 6 * N;
        out->bodies.assign(static_cast<size_t>(N), hc_h5data::Body{});
        for (int b = 0; b < N; ++b) {
            hc_h5data::Body& q     = out->bodies[static_cast<size_t>(b)];
            const std::string body = "body" + std::to_string(b + 1);
            q.disp_vol       = read_scalar(f.id, body + "/properties/disp_vol");
            const auto cg    = read_doubles(f.id, body + "/properties/cg");
            const auto cb    = read_doubles(f.id, body + "/properties/cb");
            if (cg.size() < 3 || cb.size() < 3) throw H5Error(body + ": cg/cb must have 3 entries");
            const auto lin = read_doubles(f.id, body + "/hydro_coeffs/linear_restoring_stiffness");
            if (lin.size() != 36) throw H5Error(body + ": linear_restoring_stiffness must be 6x6");
            for (int k = 0; k < 3; ++k) {
                q.cg[k] = cg[static_cast<size_t>(k)];
                q.cb[k] = cb[static_cast<size_t>(k)];
            }
            for (int k = 0; k < 36; ++k) q.lin[k] = lin[static_cast<size_t>(k)];
            q.ainf = read_doubles(f.id, body + "/hydro_coeffs/added_mass/inf_freq");
            if (q.ainf.size() != static_cast<size_t>(6) * D) throw H5Error(body + ": added_mass/inf_freq must be 6 x 6N");
            q.rirf_t = read_doubles(f.id, body + "/hydro_coeffs/radiation_damping/impulse_response_fun/t");
            std::vector<hsize_t> kd;
            q.K = read_doubles(f.id, body + "/hydro_coeffs/radiation_damping/impulse_response_fun/K", &kd);
            if (kd.size() != 3 || kd[0] != 6 || kd[1] != static_cast<hsize_t>(D) || kd[2] != q.rirf_t.size())
                throw H5Error(body + ": impulse_response_fun/K must be {6, 6N, len(t)}");
            q.mag   = read_doubles(f.id, body + "/hydro_coeffs/excitation/mag");
            q.phase = read_doubles(f.id, body + "/hydro_coeffs/excitation/phase");
            if (q.mag.size() != 6 * nw || q.phase.size() != 6 * nw) throw H5Error(body + ": excitation mag/phase must be {6,1,len(w)}");
            q.exc_t = read_doubles(f.id, body + "/hydro_coeffs/excitation/impulse_response_fun/t");
            q.exc_f = read_doubles(f.id, body + "/hydro_coeffs/excitation/impulse_response_fun/f");
            if (q.exc_f.size() != 6 * q.exc_t.size()) throw H5Error(body + ": excitation impulse_response_fun/f must be {6,1,len(t)}");
        }
    } catch (const std::exception& e) {
        if (err && errlen) std::snprintf(err, errlen, "%s", e.what());
        return HC_ERR_RUNTIME;
    }
    return HC_OK;
}

extern "C" int hc_bemio_load(hc_ctx* ctx, const char* path, char* err, size_t errlen) {
    try {
        int N = 0;
        ok(ctx, hc_get_sizes(ctx, &N, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr));
        hc_h5data d;
        const int rc = hc_bemio_read(path, N, &d, err, errlen);
        if (rc != HC_OK) return rc;
        ok(ctx, hc_set_simulation_parameters(ctx, d.rho, d.g, d.water_depth));
        for (int b = 0; b < N; ++b) {
            const hc_h5data::Body& q = d.bodies[static_cast<size_t>(b)];
            ok(ctx, hc_set_body_properties(ctx, b, q.disp_vol, q.cg, q.cb));
            ok(ctx, hc_set_hydrostatic_stiffness(ctx, b, q.lin));
            ok(ctx, hc_set_added_mass_inf(ctx, b, q.ainf.data()));
            ok(ctx, hc_set_rirf(ctx, b, q.rirf_t.data(), static_cast<int>(q.rirf_t.size()), q.K.data()));
            ok(ctx, hc_set_excitation_rao(ctx, b, d.w.data(), static_cast<int>(d.w.size()), q.mag.data(), q.phase.data()));
            ok(ctx, hc_set_excitation_irf(ctx, b, q.exc_t.data(), static_cast<int>(q.exc_t.size()), q.exc_f.data()));
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

extern "C" int hc_bemio_export_irregular(hc_ctx* ctx, const char* path, char* err, size_t errlen) {
    try {
        int nf = 0, nt = 0;
        ok(ctx, hc_get_sizes(ctx, nullptr, nullptr, nullptr, nullptr, &nf, &nt, nullptr, nullptr));
        std::vector<double> f(nf), S(nf), t(nt), eta(nt);
        if (nf) ok(ctx, hc_get_spectrum(ctx, f.data(), S.data(), nullptr, nullptr, nullptr));
        if (nt) ok(ctx, hc_get_eta_table(ctx, t.data(), eta.data()));
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
