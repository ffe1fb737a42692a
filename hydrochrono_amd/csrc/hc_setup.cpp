// hc_setup.cpp -- lifecycle, ingest (H5FileInfo::ReadH5Data scaling rules), hc_finalize, wave models and configuration, synthetic
// many-body inputs: the init-time half of the C ABI.
#include "hc_internal.hpp"
#include "hc_h5data.hpp"

using namespace hc::detail;

namespace {
// init-time stopwatches (hc_init_stats): host wall clock, and HIP events around a kernel on the context's stream
double seconds_since(const std::chrono::steady_clock::time_point& t0) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
struct KernelTimer {
    hipEvent_t a = nullptr, b = nullptr;
    hipStream_t s;
    explicit KernelTimer(hipStream_t stream) : s(stream) {
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { a = b = nullptr; (void)hipGetLastError(); return; }
        (void)hipEventRecord(a, s);
    }
    double stop() {  // seconds between construction and now on the stream (waits for the stream)
        float ms = 0.0f;
        if (a && b && hipEventRecord(b, s) == hipSuccess && hipEventSynchronize(b) == hipSuccess) (void)hipEventElapsedTime(&ms, a, b);
        else (void)hipStreamSynchronize(s);
        (void)hipGetLastError();
        return 1e-3 * static_cast<double>(ms);
    }
    ~KernelTimer() {
        if (a) (void)hipEventDestroy(a);
        if (b) (void)hipEventDestroy(b);
    }
};

// CreateSpectrum (src/wave_types.cpp:643-676): frequencies, PM/JONSWAP densities, trapezoid widths, mt19937 phases,
// wavenumbers, plus the component amplitude sqrt(2 S df) and angular frequency of GetEtaIrregular (:39-40).
struct Spectrum {
    int nf = 0;
    std::vector<double> f, S, df, phase, k, amp, omega;
};

Spectrum build_spectrum(const hc_ctx* c, const hc_irregular_wave_params& p) {
    Spectrum sp;
    if (p.nfrequencies == 0) {
        const double df = 1.0 / p.simulation_duration;
        sp.nf           = static_cast<int>(std::ceil((p.frequency_max - p.frequency_min) / df));
    } else {
        sp.nf = static_cast<int>(p.nfrequencies);
    }
    require(sp.nf >= 1, HC_ERR_INVALID, "no wave components");
    sp.f = hc::linspaced(sp.nf, p.frequency_min, p.frequency_max);
    std::sort(sp.f.begin(), sp.f.end());  // PiersonMoskowitzSpectrumHz sorts its argument in place (:681)
    sp.S     = hc::jonswap_spectrum_hz(sp.f, p.wave_height, p.wave_period, p.peak_enhancement_factor, p.is_normalized != 0);
    sp.df    = hc::trapezoid_widths(sp.f);
    sp.phase = hc::random_phases(sp.nf, p.seed);
    sp.k.resize(sp.nf);
    sp.amp.resize(sp.nf);
    sp.omega.resize(sp.nf);
    const double two_pi = 2 * M_PI;
    for (int i = 0; i < sp.nf; ++i) {
        sp.k[i]     = hc::wave_number(two_pi * sp.f[i], c->depth, c->g);
        sp.amp[i]   = std::sqrt(2 * sp.S[i] * sp.df[i]);
        sp.omega[i] = two_pi * sp.f[i];
    }
    return sp;
}

}  // namespace

// =================================================================================================
extern "C" {


const char* hc_version(void) { return kVersion; }

int hc_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int hc_device_local_cpus(int device_id, char* out, size_t out_bytes) {
    if (!out || out_bytes == 0) return HC_ERR_INVALID;
    const std::string cpus = device_local_cpus(device_id);
    std::snprintf(out, out_bytes, "%s", cpus.c_str());
    return cpus.empty() ? HC_ERR_UNSUPPORTED : HC_OK;
}

int hc_bind_thread_to_device(int device_id) { return bind_calling_thread_to_device(device_id) ? HC_OK : HC_ERR_UNSUPPORTED; }

const char* hc_last_error(const hc_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int hc_create_sharded(int num_bodies, int body_begin, int body_end, int device_id, hc_ctx** out) {
    if (!out) return HC_ERR_INVALID;
    *out = nullptr;
    try {
        require(num_bodies > 0, HC_ERR_INVALID, "num_bodies must be positive");
        require(body_begin >= 0 && body_begin < body_end && body_end <= num_bodies, HC_ERR_INVALID, "invalid body shard");
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
            throw Error(HC_ERR_DEVICE, "no HIP device available: the hydro-force path has no CPU fallback");
        require(device_id >= 0 && device_id < ndev, HC_ERR_DEVICE, "device_id out of range");
        HC_HIP(hipSetDevice(device_id));
        hipDeviceProp_t prop;
        HC_HIP(hipGetDeviceProperties(&prop, device_id));
        if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
            throw Error(HC_ERR_DEVICE, std::string("device is ") + prop.gcnArchName + ", kernels are built for gfx950 only");
        std::unique_ptr<hc_ctx> c(new hc_ctx);
        c->N      = num_bodies;
        c->b0     = body_begin;
        c->b1     = body_end;
        c->nloc   = body_end - body_begin;
        c->D      = 6 * num_bodies;
        c->Dloc   = 6 * c->nloc;
        c->device  = device_id;
        c->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        c->bodies.resize(num_bodies);
        HC_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        HC_HIP(hipStreamCreateWithFlags(&c->stream_am, hipStreamNonBlocking));
        HC_HIP(hipEventCreateWithFlags(&c->ev_fin, hipEventDisableTiming));
        HC_HIP(hipEventCreateWithFlags(&c->ev_bg, hipEventDisableTiming));
        hc_tapered_direct_options_default(&c->taper);
        hc_irregular_wave_params_default(&c->irr);
        count_context_on_device(device_id, +1);
        *out = c.release();
    } catch (const Error& e) {
        g_create_error = e.what();
        return e.status;
    } catch (const std::exception& e) {
        g_create_error = e.what();
        return HC_ERR_RUNTIME;
    }
    return HC_OK;
}

int hc_create(int num_bodies, int device_id, hc_ctx** out) { return hc_create_sharded(num_bodies, 0, num_bodies, device_id, out); }

void hc_destroy(hc_ctx* ctx) {
    if (!ctx) return;
    count_context_on_device(ctx->device, -1);
    (void)hipSetDevice(ctx->device);
    delete ctx->dq;  // drains its queue
    ctx->dq = nullptr;
    (void)hipDeviceSynchronize();  // steps may still be running on a caller's stream; the buffers go away below
    if (ctx->ext_tag_host) (void)hipHostUnregister(ctx->ext_tag_host);
    for (auto& ev : ctx->events) {
        (void)hipEventDestroy(ev.a);
        (void)hipEventDestroy(ev.b);
    }
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    if (ctx->stream_am) (void)hipStreamDestroy(ctx->stream_am);
    if (ctx->ev_fin) (void)hipEventDestroy(ctx->ev_fin);
    if (ctx->ev_bg) (void)hipEventDestroy(ctx->ev_bg);
    delete ctx;
}

// ---- ingest -----------------------------------------------------------------------------------
int hc_set_simulation_parameters(hc_ctx* c, double rho, double g, double water_depth) {
    HC_API_BEGIN(c)
    require(!c->finalized, HC_ERR_INVALID, "context already finalized");
    c->rho = rho;
    c->g = g;
    c->depth = water_depth;
    c->have_sim = true;
    HC_API_END(c)
}

int hc_set_body_properties(hc_ctx* c, int body, double disp_vol, const double cg[3], const double cb[3]) {
    HC_API_BEGIN(c)
    check_body(c, body);
    require(cg && cb, HC_ERR_INVALID, "null pointer");
    auto& b = c->bodies[body];
    b.disp_vol = disp_vol;
    std::copy(cg, cg + 3, b.cg);
    std::copy(cb, cb + 3, b.cb);
    b.have_props = true;
    HC_API_END(c)
}

int hc_set_hydrostatic_stiffness(hc_ctx* c, int body, const double lin[36]) {
    HC_API_BEGIN(c)
    check_body(c, body);
    require(lin, HC_ERR_INVALID, "null pointer");
    std::copy(lin, lin + 36, c->bodies[body].lin);
    c->bodies[body].have_lin = true;
    HC_API_END(c)
}

int hc_set_added_mass_inf(hc_ctx* c, int body, const double* A) {
    HC_API_BEGIN(c)
    check_body(c, body);
    require(A, HC_ERR_INVALID, "null pointer");
    require(c->have_sim, HC_ERR_INVALID, "set simulation parameters first (rho scales the added mass)");
    auto& b = c->bodies[body];
    b.have_ainf = true;
    if (is_local(c, body)) {
        b.ainf.assign(A, A + static_cast<size_t>(6) * c->D);
        for (auto& x : b.ainf) x *= c->rho;  // src/h5fileinfo.cpp:61
    }
    HC_API_END(c)
}

namespace {
// (K may be null for a body this context does not own: only the time vector of such a body is looked at)
void set_rirf(hc_ctx* c, int body, const double* t, int S, const double* K) {
    check_body(c, body);
    require(t && S > 0 && (K || !is_local(c, body)), HC_ERR_INVALID, "null pointer or empty IRF");
    require(c->have_sim, HC_ERR_INVALID, "set simulation parameters first (rho scales the radiation IRF)");
    require(!c->finalized, HC_ERR_INVALID, "context already finalized");
    if (c->S == 0) {
        c->S = S;
        c->tau.assign(t, t + S);
        setup_panel_geometry(c);
        c->dK.alloc(hc::panel_doubles(c->ntiles, c->ngp));
        HC_HIP(hipMemsetAsync(c->dK.p, 0, c->dK.n * sizeof(double), c->stream));  // padding rows / columns stay zero
        c->d_stage.alloc(static_cast<size_t>(6) * c->D * S);
    } else {
        require(S == c->S, HC_ERR_RUNTIME, "RIRF time vectors have to be exactly the same for all bodies (length differs)");
        for (int j = 0; j < S; ++j)
            if (std::fabs(t[j] - c->tau[j]) > 1e-10)
                throw Error(HC_ERR_RUNTIME, "RIRF time vectors have to be exactly the same for all bodies.");
    }
    c->bodies[body].have_rirf = true;
    for (int j = 0; j < S; ++j)
        if (t[j] < 0.0) c->device_errors_possible = true;  // a query t - tau could then exceed t (reference: throws at :370)
    if (is_local(c, body)) {
        const double bytes = static_cast<double>(6) * c->D * S * sizeof(double);
        const auto t0 = std::chrono::steady_clock::now();
        HC_HIP(hipMemcpyAsync(c->d_stage.p, K, static_cast<size_t>(6) * c->D * S * sizeof(double), hipMemcpyHostToDevice, c->stream));
        HC_HIP(hipStreamSynchronize(c->stream));
        c->init.rirf_h2d_seconds += seconds_since(t0);
        c->init.rirf_h2d_bytes += bytes;
        {
            KernelTimer kt(c->stream);
            hc::launch_relayout_rirf(c->d_stage.p, c->dK.p, c->ngp, c->D, S, 6 * (body - c->b0), c->rho, c->stream);
            HC_HIP(hipGetLastError());
            c->init.rirf_relayout_seconds += kt.stop();
            c->init.rirf_relayout_bytes += 2.0 * bytes;
        }
        HC_HIP(hipStreamSynchronize(c->stream));
        c->proc_ready = false;
    }
}
}  // namespace

int hc_set_rirf(hc_ctx* c, int body, const double* t, int S, const double* K) {
    HC_API_BEGIN(c)
    require(K != nullptr, HC_ERR_INVALID, "null pointer or empty IRF");
    set_rirf(c, body, t, S, K);
    HC_API_END(c)
}

int hc_set_excitation_rao(hc_ctx* c, int body, const double* w, int nw, const double* mag, const double* phase) {
    HC_API_BEGIN(c)
    check_body(c, body);
    require(w && mag && phase && nw > 0, HC_ERR_INVALID, "null pointer or empty RAO");
    require(c->have_sim, HC_ERR_INVALID, "set simulation parameters first (rho*g scales the excitation magnitude)");
    auto& b = c->bodies[body];
    b.rao_w.assign(w, w + nw);
    b.rao_mag.assign(mag, mag + static_cast<size_t>(6) * nw);
    const double rg = c->rho * c->g;
    for (auto& x : b.rao_mag) x = x * rg;  // src/h5fileinfo.cpp:73-75
    b.rao_phase.assign(phase, phase + static_cast<size_t>(6) * nw);
    b.have_rao = true;
    HC_API_END(c)
}

int hc_set_excitation_irf(hc_ctx* c, int body, const double* t, int n, const double* f) {
    HC_API_BEGIN(c)
    check_body(c, body);
    require(t && f && n > 0, HC_ERR_INVALID, "null pointer or empty excitation IRF");
    require(c->have_sim, HC_ERR_INVALID, "set simulation parameters first (rho*g scales the excitation IRF)");
    auto& b = c->bodies[body];
    b.exirf_t.assign(t, t + n);
    b.have_exirf = true;
    if (is_local(c, body)) {
        b.exirf_f.assign(f, f + static_cast<size_t>(6) * n);
        const double rg = c->rho * c->g;
        for (auto& x : b.exirf_f) x *= rg;  // src/h5fileinfo.cpp:90
    }
    HC_API_END(c)
}

namespace {
// The HDF5 code lives in libhc_bemio.so (built only where libhdf5 exists) next to this library.
void* bemio_symbol(const char* name) {
    Dl_info info;
    std::string dir = ".";
    if (dladdr(reinterpret_cast<void*>(&hc_version), &info) && info.dli_fname) {
        std::string full(info.dli_fname);
        const size_t slash = full.find_last_of('/');
        if (slash != std::string::npos) dir = full.substr(0, slash);
    }
    const std::string lib = dir + "/libhc_bemio.so";
    void* h = dlopen(lib.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!h) throw Error(HC_ERR_UNSUPPORTED, std::string("HDF5 support not available: ") + dlerror());
    void* fn = dlsym(h, name);
    if (!fn) throw Error(HC_ERR_UNSUPPORTED, std::string("libhc_bemio.so lacks ") + name);
    return fn;
}
}  // namespace

// ---- the file without a device context (include/hydroc_amd/h5fileinfo.h) ----
extern "C++" {
namespace {
template <class F>
int h5_guard(F&& f) {
    try {
        f();
    } catch (const Error& e) {
        g_create_error = e.what();
        return e.status;
    } catch (const std::exception& e) {
        g_create_error = e.what();
        return HC_ERR_RUNTIME;
    }
    return HC_OK;
}
const hc_h5data::Body& h5_body(const hc_h5data* d, int body) {
    require(d != nullptr, HC_ERR_INVALID, "null data");
    if (body < 0 || body >= static_cast<int>(d->bodies.size())) throw Error(HC_ERR_OUT_OF_RANGE, "body index out of range");
    return d->bodies[static_cast<size_t>(body)];
}
void copy_out(double* dst, const std::vector<double>& src, double scale = 1.0) {
    if (!dst) return;
    for (size_t k = 0; k < src.size(); ++k) dst[k] = src[k] * scale;
}
}  // namespace
}  // extern "C++"

int hc_h5_read(const char* path, int num_bodies, hc_h5data** out) {
    if (!out) return HC_ERR_INVALID;
    *out = nullptr;
    return h5_guard([&] {
        require(path, HC_ERR_INVALID, "null path");
        require(num_bodies > 0, HC_ERR_INVALID, "num_bodies must be positive");
        using read_fn_t = int (*)(const char*, int, hc_h5data*, char*, size_t);
        const read_fn_t fn = reinterpret_cast<read_fn_t>(bemio_symbol("hc_bemio_read"));
        std::unique_ptr<hc_h5data> d(new hc_h5data);
        char msg[1024] = {0};
        const int rc = fn(path, num_bodies, d.get(), msg, sizeof msg);
        if (rc != HC_OK) throw Error(rc, msg[0] ? std::string(msg) : std::string("cannot read ") + path);
        *out = d.release();
    });
}

void hc_h5_free(hc_h5data* data) { delete data; }

int hc_h5_get_sizes(const hc_h5data* d, int* num_bodies, double* rho, double* g, double* water_depth, int body, int* S, int* nw, int* L) {
    return h5_guard([&] {
        require(d != nullptr, HC_ERR_INVALID, "null data");
        if (num_bodies) *num_bodies = static_cast<int>(d->bodies.size());
        if (rho) *rho = d->rho;
        if (g) *g = d->g;
        if (water_depth) *water_depth = d->water_depth;
        if (S || nw || L) {
            const hc_h5data::Body& q = h5_body(d, body);
            if (S) *S = static_cast<int>(q.rirf_t.size());
            if (nw) *nw = static_cast<int>(d->w.size());
            if (L) *L = static_cast<int>(q.exc_t.size());
        }
    });
}

int hc_h5_get_body(const hc_h5data* d, int body, double* disp_vol, double cg[3], double cb[3], double lin[36], double* ainf_6xD, double* rirf_t_S) {
    return h5_guard([&] {
        const hc_h5data::Body& q = h5_body(d, body);
        if (disp_vol) *disp_vol = q.disp_vol;
        for (int k = 0; k < 3; ++k) {
            if (cg) cg[k] = q.cg[k];
            if (cb) cb[k] = q.cb[k];
        }
        if (lin) std::copy(q.lin, q.lin + 36, lin);
        copy_out(ainf_6xD, q.ainf, d->rho);  // src/h5fileinfo.cpp:60-61
        copy_out(rirf_t_S, q.rirf_t);
    });
}

int hc_h5_get_rirf(const hc_h5data* d, int body, double* K) {
    return h5_guard([&] { copy_out(K, h5_body(d, body).K); });
}

int hc_h5_get_excitation_rao(const hc_h5data* d, int body, double* w, double* mag, double* phase) {
    return h5_guard([&] {
        const hc_h5data::Body& q = h5_body(d, body);
        copy_out(w, d->w);
        copy_out(mag, q.mag, d->rho * d->g);  // :73-75
        copy_out(phase, q.phase);
    });
}

int hc_h5_get_excitation_irf(const hc_h5data* d, int body, double* t, double* f) {
    return h5_guard([&] {
        const hc_h5data::Body& q = h5_body(d, body);
        copy_out(t, q.exc_t);
        copy_out(f, q.exc_f, d->rho * d->g);  // :89-90
    });
}

// Body by body: the reader hands over one body's datasets at a time and they are released before the next is read, so
// the host holds at most one {6, 6N, S} tensor -- not all N of them ((6N)^2 S doubles) -- and a row-shard context does not read the
// tensors of the bodies it does not own at all (every shard context of a system reads the file).
int hc_load_bemio_h5(hc_ctx* c, const char* path) {
    HC_API_BEGIN(c)
    require(path, HC_ERR_INVALID, "null path");
    using read_fn_t = int (*)(const char*, int, int, int, hc_h5data*, char*, size_t);
    const read_fn_t read_body = reinterpret_cast<read_fn_t>(bemio_symbol("hc_bemio_read_body"));
    auto chk = [&](int rc) {
        if (rc != HC_OK) throw Error(rc, c->err);
    };
    for (int b = -1; b < c->N; ++b) {
        hc_h5data d;
        char msg[1024] = {0};
        const auto t_read = std::chrono::steady_clock::now();
        const int rc = read_body(path, c->N, b, (b >= 0 && is_local(c, b)) ? 1 : 0, &d, msg, sizeof msg);
        if (rc != HC_OK) throw Error(rc, msg[0] ? std::string(msg) : std::string("cannot read ") + path);
        c->init.h5_read_seconds += seconds_since(t_read);
        if (b >= 0 && !d.bodies.empty()) {
            const hc_h5data::Body& qb = d.bodies[0];
            c->init.h5_read_bytes += 8.0 * static_cast<double>(qb.K.size() + qb.ainf.size() + qb.rirf_t.size() + qb.mag.size() + qb.phase.size() + qb.exc_t.size() +
                                                               qb.exc_f.size() + d.w.size() + 43);
        }
        if (b < 0) {
            chk(hc_set_simulation_parameters(c, d.rho, d.g, d.water_depth));
            continue;
        }
        const hc_h5data::Body& q = d.bodies[0];
        chk(hc_set_body_properties(c, b, q.disp_vol, q.cg, q.cb));
        chk(hc_set_hydrostatic_stiffness(c, b, q.lin));
        chk(hc_set_added_mass_inf(c, b, q.ainf.data()));
        set_rirf(c, b, q.rirf_t.data(), static_cast<int>(q.rirf_t.size()), q.K.empty() ? nullptr : q.K.data());
        chk(hc_set_excitation_rao(c, b, d.w.data(), static_cast<int>(d.w.size()), q.mag.data(), q.phase.data()));
        chk(hc_set_excitation_irf(c, b, q.exc_t.data(), static_cast<int>(q.exc_t.size()), q.exc_f.data()));
    }
    HC_API_END(c)
}

int hc_export_irregular_inputs_h5(hc_ctx* c, const char* path) {
    HC_API_BEGIN(c)
    require(path, HC_ERR_INVALID, "null path");
    require(c->wave_kind == hc::kWaveIrregular || c->wave_kind == hc::kWaveSpectral, HC_ERR_INVALID, "no irregular wave model attached");
    using export_fn_t = int (*)(const char*, const double*, const double*, int, const double*, const double*, int, char*, size_t);
    const export_fn_t fn = reinterpret_cast<export_fn_t>(bemio_symbol("hc_bemio_export_irregular"));
    // what SimulationExporter::WriteIrregularInputs writes (src/simulation_exporter.cpp:365-393): spectrum and free-surface table
    char msg[1024] = {0};
    const int rc = fn(path, c->spec_f.data(), c->spec_S.data(), static_cast<int>(c->spec_f.size()), c->eta_t.data(), c->eta.data(),
                      static_cast<int>(c->eta_t.size()), msg, sizeof msg);
    if (rc != HC_OK) throw Error(rc, msg[0] ? std::string(msg) : c->err);
    HC_API_END(c)
}

int hc_finalize(hc_ctx* c) {
    HC_API_BEGIN(c)
    require(!c->finalized, HC_ERR_INVALID, "context already finalized");
    require(c->have_sim, HC_ERR_INVALID, "simulation parameters missing");
    require(c->S > 0, HC_ERR_INVALID, "radiation IRF missing");
    const auto t_finalize = std::chrono::steady_clock::now();
    for (int b = c->b0; b < c->b1; ++b) {
        const auto& bd = c->bodies[b];
        require(bd.have_props && bd.have_lin && bd.have_ainf && bd.have_rirf, HC_ERR_INVALID,
                "a local body lacks properties, hydrostatic stiffness, added mass or radiation IRF");
    }
    // trapezoid widths (src/hydro_forces.cpp:181-190)
    c->width = hc::trapezoid_widths(c->tau);
    c->d_tau.upload(c->tau, c->stream);
    c->d_width.upload(c->width, c->stream);
    // hydrostatics tables; equilibrium = cg, cb - cg (:208-216)
    std::vector<double> lin(static_cast<size_t>(c->nloc) * 36), cg(static_cast<size_t>(c->nloc) * 3), cbm(static_cast<size_t>(c->nloc) * 3),
        vol(c->nloc);
    for (int bl = 0; bl < c->nloc; ++bl) {
        const auto& bd = c->bodies[c->b0 + bl];
        std::copy(bd.lin, bd.lin + 36, lin.begin() + static_cast<size_t>(bl) * 36);
        for (int k = 0; k < 3; ++k) {
            cg[bl * 3 + k]  = bd.cg[k];
            cbm[bl * 3 + k] = bd.cb[k] - bd.cg[k];
        }
        vol[bl] = bd.disp_vol;
    }
    c->d_lin.upload(lin, c->stream);
    c->d_cg.upload(cg, c->stream);
    c->d_cbmcg.upload(cbm, c->stream);
    c->d_vol.upload(vol, c->stream);
    // added mass rows (src/chloadaddedmass.cpp:18-21)
    c->ainf_host.assign(static_cast<size_t>(c->Dloc) * c->D, 0.0);
    for (int bl = 0; bl < c->nloc; ++bl) {
        const auto& bd = c->bodies[c->b0 + bl];
        std::copy(bd.ainf.begin(), bd.ainf.end(), c->ainf_host.begin() + static_cast<size_t>(bl) * 6 * c->D);
    }
    c->d_ainf.upload(c->ainf_host, c->stream);
    c->d_vec_w.alloc(c->D);
    c->d_vec_R.alloc(c->Dloc);
    c->d_stage.release();
    // history ring
    ring_alloc(c, std::max(64, c->S + 2) + hc::kRewindSlack);
    c->times.clear();
    c->retired.clear();
    c->have_prev = c->have_prev_device = false;
    c->prev_time = c->prev_time_device = -1.0;
    {
        const int want = HC_TUNE_INT("HC_LOOKAHEAD", 32);
        c->lookahead   = want <= 0 ? 0 : (want <= 16 ? 16 : hc::kDepthDefault);
        c->mt_block64  = HC_TUNE_INT("HC_BLOCK64_MT", 3);
        if (c->mt_block64 != 3 && c->mt_block64 != 4 && c->mt_block64 != 6) c->mt_block64 = 3;
        c->pass_ahead  = default_pass_ahead(c);  // hc_set_pass_schedule
        c->pass_slices = default_pass_slices(c);
        // the gap beyond which "one block ahead" pays (profiles/r05/ahead_probe_fine_gaps.txt): C3 -- 0 - 3 us: the schedules within
        // noise of each other (17.4 / 18.6 / 17.4 / 16.9 at block start, 18.8 / 18.6 / 16.8 / 17.6 ahead), 5 us: 18.4 against 17.0,
        // 8 us: 18.1 against 14.8, 20 us: 16.8 against 11.9 us per step; a C4/8 rank -- ahead wins at every gap, back to back included
        // (70.1 against 68.6 us, worst step 1.58 against 0.33 ms; 10 us: 70.2 against 60.9): the kernels of its step path are latency-bound
        // and run beside the pass for free.  Larger slices of a wide system (profiles/r05/ahead_probe_shard_sizes.txt: rows of 128 / 256 /
        // 512 of 512 bodies, K slices of 19 / 39 / 77 GB) have bandwidth-bound step kernels that only share the memory with a sliced pass:
        // back to back the pass at block start wins by 4-6 %, "ahead" from gaps of 10-20 / 10-20 / 50-100 us on -- about a tenth of what the
        // pass costs per step.  So: wide and at most 12 GB of K in this context: 0; wide and larger: a tenth of (K bytes / 32 steps / 6.3 TB/s);
        // otherwise 4 us.
        {
            const double k_local = 8.0 * static_cast<double>(c->Dloc) * c->D * c->S;
            int def_us = 4;
            if (hc::near_slices_for(c->D) > 1) def_us = k_local <= 12e9 ? 0 : static_cast<int>(0.1 * k_local / hc::kDepthDefault / 6.3e12 * 1e6 + 0.5);
            c->gap_threshold = 1e-6 * std::max(0, env_int("HC_PASS_AHEAD_GAP_US", def_us));
        }
        reset_schedule_state(c);
        c->pass_concurrent = env_int("HC_PASS_CONCURRENT", 1) != 0;
        c->pass_free_cus   = std::max(1, std::min(28, HC_TUNE_INT("HC_PASS_FREE_CUS", 4)));
        c->ahead.active = false;
    }
    // GEMV scratch
    choose_conv_config(c);
    // step I/O
    c->d_state.alloc(static_cast<size_t>(12) * c->N);
    c->d_hs.alloc(c->Dloc);
    c->d_rad.alloc(c->Dloc);
    c->d_waves.alloc(c->Dloc);
    c->d_total.alloc(c->Dloc);
    for (auto* b : {&c->d_hs, &c->d_rad, &c->d_waves, &c->d_total}) HC_HIP(hipMemsetAsync(b->p, 0, b->n * sizeof(double), c->stream));
    c->d_err.alloc(3);  // [0] error flag of the convolution kernels, [1] work-item counter of the look-ahead pass, [2] ... of a pass on the pass lane
    HC_HIP(hipMemsetAsync(c->d_err.p, 0, 3 * sizeof(int), c->stream));
    c->h_state.alloc(static_cast<size_t>(2) * (12 * c->N + 1));  // two halves used alternately by hc_step, see there (+ a canary word each)
    c->bar_state.alloc(static_cast<size_t>(2) * (12 * c->N + 1));
    c->h_canary.alloc(4);
    std::memset(c->h_canary.p, 0, 4 * sizeof(unsigned long long));
    c->fault_stale_state_at = HC_TUNE_INT("HC_FAULT_STALE_STATE_AT", -1);
    if (c->bar_state.host_ok) {
        // trust, but verify: what the host stores through the BAR must be what a device-side copy sees
        const size_t nb = c->bar_state.n;
        for (size_t k = 0; k < nb; ++k) c->bar_state.p[k] = 0.5 + static_cast<double>(k);
        _mm_sfence();
        hc::DeviceBuffer<double> tmp;
        tmp.alloc(nb);
        std::vector<double> back(nb);
        HC_HIP(hipMemcpyAsync(tmp.p, c->bar_state.p, nb * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        HC_HIP(hipMemcpyAsync(back.data(), tmp.p, nb * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HC_HIP(hipStreamSynchronize(c->stream));
        for (size_t k = 0; k < nb && c->bar_state.host_ok; ++k) c->bar_state.host_ok = back[k] == 0.5 + static_cast<double>(k);
    }
    c->h_out.alloc(static_cast<size_t>(4) * c->Dloc);
    c->h_err.alloc(1);
    c->h_am.alloc(static_cast<size_t>(c->D) + c->Dloc + 1);  // w | incoming R | the product's canary word
    c->bar_am.alloc(static_cast<size_t>(c->D) + c->Dloc + 1);
    c->h_tag_am.alloc(static_cast<size_t>(2) * c->Dloc + 2);  // [Dloc][2] result granules + the canary granule
    c->bar_selftest.alloc(2);
    c->h_tag_selftest.alloc(2);
    c->d_selftest.alloc(1);
    std::memset(c->h_tag_am.p, 0, c->h_tag_am.n * sizeof(unsigned long long));
    c->seq_am = 0;
    c->h_tag.alloc(static_cast<size_t>(4) * c->Dloc);  // two halves of [Dloc][2], used alternately (step sequence parity)
    std::memset(c->h_tag.p, 0, c->h_tag.n * sizeof(unsigned long long));
    c->seq = 0;
    c->last_total.assign(c->Dloc, 0.0);
    c->d_scratch.alloc(static_cast<size_t>(4) * c->Dloc);
    c->d_zero_state.alloc(static_cast<size_t>(12) * c->N);
    HC_HIP(hipMemsetAsync(c->d_zero_state.p, 0, c->d_zero_state.n * sizeof(double), c->stream));
    c->zero_copy_max_bodies = HC_TUNE_INT("HC_ZERO_COPY_BODIES", 64);
    c->arm_mode             = std::min(2, std::max(0, env_int("HC_ARM", 1)));  // 0 never, 1 adaptive, 2 always: parking of the direct queue between steps
    // default wave model: NoWave for all bodies (the reference's default NoWave() covers one body only and is
    // read out of bounds for N > 1, src/hydro_forces.cpp:758-760; that overread is deliberately not reproduced)
    c->wave_kind   = hc::kWaveNone;
    c->wave_nb_arg = c->N;
    choose_exc_config(c);
    alloc_partials(c);
    c->prof.conv_kernel_bytes  = 8.0 * (static_cast<double>(c->Dloc) * c->S * c->D + static_cast<double>(c->S) * c->D);
    c->prof.block_kernel_bytes = hc::kDepthDefault * c->prof.conv_kernel_bytes;
    c->plan                    = hc::Plan{};
    HC_HIP(hipStreamSynchronize(c->stream));
    const auto t_direct = std::chrono::steady_clock::now();
    setup_direct(c);
    // the pass lane (its queue, CU mask and self-test: 131 synchronous dispatches) is made here, not inside the first step that
    // starts a pass one block ahead
    if (pass_ahead_possible(c) && c->lookahead > 0) (void)pass_lane_ready(c);
    c->init.direct_setup_seconds += seconds_since(t_direct);
    c->init.finalize_seconds += seconds_since(t_finalize);
    c->finalized = true;
    HC_API_END(c)
}

// The wave model is about to change: excitation rows a look-ahead pass has precomputed belong to the previous model -- those of the
// current block (Plan::has_exc) AND those of the pass one block ahead, whether its rows are complete already (the next block would adopt
// them, hc_step.cpp: plan.has_exc = ahead.has_exc) or its last slice, which carries the excitation work items with the OLD tables in
// its arguments, is still to be issued.  The radiation rows stay valid.  (Found by profiles/fuzz_parity.py, seed 40: a wave model
// redrawn under "one block ahead" gave the next block the old model's excitation force.)
static void drop_lookahead_excitation(hc_ctx* c) {
    c->plan.has_exc        = false;
    c->ahead.has_exc       = false;
    c->ahead.exc_once      = 0.0;
    c->ahead.args.nchunks_ex = 0;
}

// ---- configuration ----------------------------------------------------------------------------
int hc_set_gravity(hc_ctx* c, const double g[3]) {
    HC_API_BEGIN(c)
    require(g, HC_ERR_INVALID, "null pointer");
    std::copy(g, g + 3, c->gsys);
    HC_API_END(c)
}

int hc_set_wave_none(hc_ctx* c, int num_bodies_arg) {
    HC_API_BEGIN(c)
    drop_lookahead_excitation(c);
    HC_HIP(hipDeviceSynchronize());  // the tables replaced below may be in use by steps on a caller's stream
    require(c->finalized, HC_ERR_INVALID, "hc_finalize has not been called");
    require(num_bodies_arg >= 0, HC_ERR_INVALID, "negative body count");
    c->wave_kind   = hc::kWaveNone;
    c->wave_nb_arg = num_bodies_arg;
    choose_exc_config(c);
    HC_API_END(c)
}

int hc_set_wave_regular(hc_ctx* c, int num_bodies_arg, double amplitude, double omega) {
    HC_API_BEGIN(c)
    drop_lookahead_excitation(c);
    HC_HIP(hipDeviceSynchronize());  // the tables replaced below may be in use by steps on a caller's stream
    require(c->finalized, HC_ERR_INVALID, "hc_finalize has not been called");
    require(num_bodies_arg >= 1 && num_bodies_arg <= c->N, HC_ERR_OUT_OF_RANGE, "regular wave created for more bodies than the hydro data holds");
    for (int b = 0; b < num_bodies_arg; ++b) require(c->bodies[b].have_rao, HC_ERR_INVALID, "excitation RAO missing for a body");
    // RegularWave::AddH5Data (src/wave_types.cpp:278-299), GetOmegaDelta (:329-333), Get*Interp (:335-352)
    const auto& w0        = c->bodies[0].rao_w;
    const double nfreq    = static_cast<double>(w0.size());
    const double dw       = w0.back() / nfreq;
    const double idx_des  = (omega / dw) - 1;
    const double frac     = idx_des - std::floor(idx_des);
    const int k0          = static_cast<int>(std::floor(idx_des));
    std::vector<double> mag(c->D, 0.0), ph(c->D, 0.0);
    for (int b = 0; b < num_bodies_arg; ++b) {
        const auto& bd = c->bodies[b];
        const int nw   = static_cast<int>(bd.rao_w.size());
        if (k0 < 0 || k0 + 1 >= nw) throw Error(HC_ERR_OUT_OF_RANGE, "regular wave frequency outside the BEM frequency list");
        for (int r = 0; r < 6; ++r) {
            const double m0 = bd.rao_mag[static_cast<size_t>(r) * nw + k0], m1 = bd.rao_mag[static_cast<size_t>(r) * nw + k0 + 1];
            const double p0 = bd.rao_phase[static_cast<size_t>(r) * nw + k0], p1 = bd.rao_phase[static_cast<size_t>(r) * nw + k0 + 1];
            mag[6 * b + r] = (frac * (m1 - m0)) + m0;
            ph[6 * b + r]  = (frac * (p1 - p0)) + p0;
        }
    }
    const double k = hc::wave_number(omega, c->depth, c->g);  // RegularWave::Initialize (:274-276)
    c->reg_mag = mag;
    c->reg_phase = ph;
    c->reg_amp = amplitude;
    c->reg_omega = omega;
    c->reg_wavenumber = k;
    std::vector<double> local(mag.begin() + 6 * c->b0, mag.begin() + 6 * c->b1);
    c->d_reg_mag.upload(local, c->stream);
    c->wave_kind   = hc::kWaveRegular;
    c->wave_nb_arg = num_bodies_arg;
    choose_exc_config(c);
    HC_API_END(c)
}

void hc_irregular_wave_params_default(hc_irregular_wave_params* p) {
    if (!p) return;
    p->num_bodies = 1;
    p->simulation_dt = 0.0;
    p->simulation_duration = 0.0;
    p->ramp_duration = 0.0;
    p->wave_height = 0.0;
    p->wave_period = 0.0;
    p->frequency_min = 0.001;
    p->frequency_max = 1.0;
    p->nfrequencies = 0;
    p->peak_enhancement_factor = 1.0;
    p->is_normalized = 0;
    p->seed = 1;
}

int hc_set_wave_irregular(hc_ctx* c, const hc_irregular_wave_params* pp) {
    HC_API_BEGIN(c)
    drop_lookahead_excitation(c);
    HC_HIP(hipDeviceSynchronize());  // the tables replaced below may be in use by steps on a caller's stream
    require(c->finalized, HC_ERR_INVALID, "hc_finalize has not been called");
    require(pp, HC_ERR_INVALID, "null parameters");
    const hc_irregular_wave_params p = *pp;
    require(p.num_bodies == c->N, HC_ERR_INVALID, "IrregularWaveParams.num_bodies_ must equal the number of hydro bodies");
    require(p.simulation_dt > 0.0, HC_ERR_INVALID, "simulation_dt must be positive");
    require(p.wave_height != 0.0 && p.wave_period != 0.0, HC_ERR_INVALID,
            "wave_height and wave_period must be non-zero (the reference leaves the free-surface table empty otherwise)");
    // Excitation-IRF time grids.  The reference keeps one grid per body (ex_irf_time_sampled_[b], src/wave_types.cpp:432-459) and
    // resamples each on its own (:572-606).  BEMIO writes one grid per file, so bodies normally share it; bodies with the same
    // grid (within 1e-10, the tolerance the reference applies to the radiation grids) form a group, and the columns of Kex are
    // the groups' resampled grids one after the other -- a body's rows are non-zero in the columns of its own group only, so
    // one launch still serves all bodies.  Groups are formed over ALL bodies (not only the local ones): the column layout, and
    // with it the summation order, is the same in every row shard of the system.
    std::vector<hc::ExGroup> groups;
    std::vector<int> group_of(c->N, -1);
    for (int b = 0; b < c->N; ++b) {
        if (!c->bodies[b].have_exirf) {
            // a row-sharded context whose caller ingested its own bodies only: the other bodies' grids are unknown here
            require(!is_local(c, b), HC_ERR_INVALID, "excitation IRF missing for a local body");
            continue;
        }
        const auto& tb = c->bodies[b].exirf_t;
        require(tb.size() >= 2, HC_ERR_INVALID, "excitation IRF with fewer than two samples");
        for (size_t g = 0; g < groups.size() && group_of[b] < 0; ++g) {
            const auto& tg = c->bodies[groups[g].first_body].exirf_t;
            bool same = tg.size() == tb.size();
            for (size_t j = 0; j < tb.size() && same; ++j) same = std::fabs(tb[j] - tg[j]) <= 1e-10;
            if (same) group_of[b] = static_cast<int>(g);
        }
        if (group_of[b] < 0) {
            hc::ExGroup g;
            g.first_body = b;
            group_of[b]  = static_cast<int>(groups.size());
            groups.push_back(g);
        }
    }
    const auto t_call = std::chrono::steady_clock::now();
    // ResampleIRF (src/wave_types.cpp:572-606), per group
    std::vector<double> ex_tau, ex_width;
    int L = 0;
    for (auto& g : groups) {
        const auto& t_old = c->bodies[g.first_body].exirf_t;
        const double t0 = t_old.front(), t1 = t_old.back();
        g.L   = static_cast<int>(std::ceil((t1 - t0) / p.simulation_dt));
        require(g.L >= 2, HC_ERR_INVALID, "excitation IRF resamples to fewer than two points");
        g.off = L;
        const std::vector<double> tg = hc::linspaced(g.L, t0, t1), wg = hc::trapezoid_widths(tg);
        g.tau_front = tg.front();
        g.tau_back  = tg.back();
        ex_tau.insert(ex_tau.end(), tg.begin(), tg.end());
        ex_width.insert(ex_width.end(), wg.begin(), wg.end());
        L += g.L;
    }
    const int Lpad = (L + 7) & ~7;
    std::vector<double> vals(static_cast<size_t>(c->Dloc) * L, 0.0);
    for (int bl = 0; bl < c->nloc; ++bl) {
        const auto& bd = c->bodies[c->b0 + bl];
        const hc::ExGroup& g = groups[group_of[c->b0 + bl]];
        const auto r = hc::resample_cubic_bspline6(bd.exirf_f, static_cast<int>(bd.exirf_t.size()), g.L);
        for (int d = 0; d < 6; ++d)
            std::copy(r.begin() + static_cast<size_t>(d) * g.L, r.begin() + static_cast<size_t>(d + 1) * g.L,
                      vals.begin() + static_cast<size_t>(6 * bl + d) * L + g.off);
    }
    c->init.wave_resample_seconds += seconds_since(t_call);
    // CreateSpectrum (:643-676)
    const auto t_spec = std::chrono::steady_clock::now();
    Spectrum sp = build_spectrum(c, p);
    c->init.wave_spectrum_seconds += seconds_since(t_spec);
    const int nf = sp.nf;
    std::vector<double>&f = sp.f, &Sd = sp.S, &dfv = sp.df, &phase = sp.phase, &kk = sp.k, &amp = sp.amp, &omg = sp.omega;
    // CreateFreeSurfaceElevation (:717-774): the min/max scan over every body's resampled grid = over the groups' ends
    double t_irf_min = 0.0, t_irf_max = 0.0;
    for (const auto& g : groups) {
        if (g.tau_front < t_irf_min) t_irf_min = g.tau_front;
        if (g.tau_front > t_irf_max) t_irf_max = g.tau_front;
        if (g.tau_back > t_irf_max) t_irf_max = g.tau_back;
        if (g.tau_back < t_irf_min) t_irf_min = g.tau_back;
    }
    const double duration = p.simulation_duration + 2 * (t_irf_max - t_irf_min);
    const int nts         = static_cast<int>(std::ceil(duration / p.simulation_dt));
    std::vector<double> eta_t = hc::linspaced(nts + 1, 0, nts * p.simulation_dt);
    for (auto& x : eta_t) x += -t_irf_max;
    const int nt = nts + 1;
    require(nt >= 2, HC_ERR_INVALID, "free-surface table too short");

    hc::DeviceBuffer<double> d_amp, d_omg, d_ph;
    const auto t_up = std::chrono::steady_clock::now();
    d_amp.upload(amp, c->stream);
    d_omg.upload(omg, c->stream);
    d_ph.upload(phase, c->stream);
    c->d_eta_t.upload(eta_t, c->stream);
    c->d_eta.alloc(nt);
    double up_s = seconds_since(t_up);
    {
        const auto t_eta = std::chrono::steady_clock::now();
        if (c->eta_mode == 1 && nf >= 2) {
            hc::eta_synthesis_fft(eta_t, amp, omg, phase, p.ramp_duration, c->d_eta_t.p, c->d_eta.p, c->stream);  // rocFFT chirp-z (host set-up + plans + kernels)
            HC_HIP(hipGetLastError());
            HC_HIP(hipStreamSynchronize(c->stream));
            c->init.wave_eta_seconds += seconds_since(t_eta);
        } else {
            KernelTimer kt(c->stream);
            hc::launch_eta_synthesis(c->d_eta_t.p, nt, d_amp.p, d_omg.p, d_ph.p, nf, p.ramp_duration, c->d_eta.p, c->stream);
            HC_HIP(hipGetLastError());
            c->init.wave_eta_seconds += kt.stop();
        }
        c->init.wave_eta_samples    = nt;
        c->init.wave_eta_components = nf;
        c->init.wave_eta_mode       = (c->eta_mode == 1 && nf >= 2) ? 1 : 0;
    }
    const auto t_up2 = std::chrono::steady_clock::now();
    std::vector<double> eta(nt);
    HC_HIP(hipMemcpyAsync(eta.data(), c->d_eta.p, nt * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HC_HIP(hipStreamSynchronize(c->stream));

    {   // excitation IRF into the same panel layout as K (row tiles of 16, column groups of 8)
        hc::DeviceBuffer<double> d_rowmajor;
        d_rowmajor.upload(vals, c->stream);
        c->ngp_ex = Lpad / 8;
        c->d_kex.alloc(hc::panel_doubles(c->ntiles, c->ngp_ex));
        HC_HIP(hipMemsetAsync(c->d_kex.p, 0, c->d_kex.n * sizeof(double), c->stream));
        hc::launch_relayout_rowmajor(d_rowmajor.p, c->Dloc, L, c->d_kex.p, c->ngp_ex, 0, c->stream);
        HC_HIP(hipGetLastError());
        HC_HIP(hipStreamSynchronize(c->stream));
    }
    c->d_ex_tau.upload(ex_tau, c->stream);
    c->d_ex_width.upload(ex_width, c->stream);
    c->init.wave_upload_seconds += up_s + seconds_since(t_up2);
    c->init.wave_upload_bytes += 8.0 * (3.0 * nf + 3.0 * nt + 2.0 * static_cast<double>(c->Dloc) * L + 2.0 * L);
    c->irr = p;
    c->L = L;
    c->Lpad = Lpad;
    c->nf = nf;
    c->nt = nt;
    c->ex_tau.swap(ex_tau);
    c->ex_width.swap(ex_width);
    c->ex_vals.swap(vals);
    c->ex_groups.swap(groups);
    c->ex_group_of.swap(group_of);
    c->spec_f.swap(f);
    c->spec_S.swap(Sd);
    c->spec_df.swap(dfv);
    c->spec_phase.swap(phase);
    c->spec_k.swap(kk);
    c->eta_t.swap(eta_t);
    c->eta.swap(eta);
    c->wave_kind   = hc::kWaveIrregular;
    c->wave_nb_arg = p.num_bodies;
    choose_exc_config(c);
    alloc_partials(c);
    c->prof.conv_kernel_bytes = 8.0 * (static_cast<double>(c->Dloc) * c->S * c->D + static_cast<double>(c->S) * c->D +
                                       static_cast<double>(c->Dloc) * L + L);
    c->prof.block_kernel_bytes = hc::kDepthDefault * 8.0 * (static_cast<double>(c->Dloc) * c->S * c->D + static_cast<double>(c->S) * c->D);
    HC_HIP(hipStreamSynchronize(c->stream));
    c->init.wave_total_seconds += seconds_since(t_call);
    HC_API_END(c)
}

int hc_set_wave_irregular_spectral(hc_ctx* c, const hc_irregular_wave_params* pp) {
    HC_API_BEGIN(c)
    drop_lookahead_excitation(c);
    HC_HIP(hipDeviceSynchronize());  // the tables replaced below may be in use by steps on a caller's stream
    require(c->finalized, HC_ERR_INVALID, "hc_finalize has not been called");
    require(pp, HC_ERR_INVALID, "null parameters");
    const hc_irregular_wave_params p = *pp;
    require(p.num_bodies == c->N, HC_ERR_INVALID, "IrregularWaveParams.num_bodies_ must equal the number of hydro bodies");
    require(p.wave_height != 0.0 && p.wave_period != 0.0, HC_ERR_INVALID, "wave_height and wave_period must be non-zero");
    for (int b = c->b0; b < c->b1; ++b) require(c->bodies[b].have_rao, HC_ERR_INVALID, "excitation RAO missing for a local body");
    Spectrum sp = build_spectrum(c, p);
    // per-component excitation RAO: RegularWave's interpolator (src/wave_types.cpp:329-352: uniform list starting at
    // d_omega, index = omega/d_omega - 1, linear between neighbours), held constant outside the BEM frequency range
    std::vector<double> mag(static_cast<size_t>(c->Dloc) * sp.nf), ph(static_cast<size_t>(c->Dloc) * sp.nf);
    for (int bl = 0; bl < c->nloc; ++bl) {
        const auto& bd  = c->bodies[c->b0 + bl];
        const int nw    = static_cast<int>(bd.rao_w.size());
        const double dw = bd.rao_w.back() / static_cast<double>(nw);
        for (int i = 0; i < sp.nf; ++i) {
            double idx = sp.omega[i] / dw - 1;
            idx        = std::min(std::max(idx, 0.0), static_cast<double>(nw - 1));
            const int k0 = std::min(static_cast<int>(std::floor(idx)), nw - 2 >= 0 ? nw - 2 : 0);
            const int k1 = std::min(k0 + 1, nw - 1);
            const double fr = idx - k0;
            for (int r = 0; r < 6; ++r) {
                const double m0 = bd.rao_mag[static_cast<size_t>(r) * nw + k0], m1 = bd.rao_mag[static_cast<size_t>(r) * nw + k1];
                const double p0 = bd.rao_phase[static_cast<size_t>(r) * nw + k0], p1 = bd.rao_phase[static_cast<size_t>(r) * nw + k1];
                mag[static_cast<size_t>(6 * bl + r) * sp.nf + i] = (fr * (m1 - m0)) + m0;
                ph[static_cast<size_t>(6 * bl + r) * sp.nf + i]  = (fr * (p1 - p0)) + p0;
            }
        }
    }
    c->d_spec_mag.upload(mag, c->stream);
    c->d_spec_phase.upload(ph, c->stream);
    c->d_spec_amp.upload(sp.amp, c->stream);
    c->d_spec_omega.upload(sp.omega, c->stream);
    c->d_spec_phi.upload(sp.phase, c->stream);
    c->irr = p;
    c->nf  = sp.nf;
    c->L = c->Lpad = c->nt = 0;
    c->spec_f.swap(sp.f);
    c->spec_S.swap(sp.S);
    c->spec_df.swap(sp.df);
    c->spec_phase.swap(sp.phase);
    c->spec_k.swap(sp.k);
    c->wave_kind   = hc::kWaveSpectral;
    c->wave_nb_arg = p.num_bodies;
    choose_exc_config(c);
    HC_API_END(c)
}

int hc_set_eta_synthesis(hc_ctx* c, int mode) {
    HC_API_BEGIN(c)
    require(mode == 0 || mode == 1, HC_ERR_INVALID, "mode must be 0 (direct FP64 sum) or 1 (rocFFT chirp-z)");
    c->eta_mode = mode;
    HC_API_END(c)
}

int hc_set_convolution_mode(hc_ctx* c, int mode) {
    HC_API_BEGIN(c)
    require(mode == 0 || mode == 1, HC_ERR_INVALID, "mode must be 0 (Baseline) or 1 (TaperedDirect)");
    c->conv_mode  = mode;
    c->plan.valid = false;
    HC_API_END(c)
}

void hc_tapered_direct_options_default(hc_tapered_direct_options* o) {
    if (!o) return;
    o->smoothing = 0;
    o->window_length = 5;
    o->rirf_end_time = -1.0;
    o->taper_start_percent = 0.8;
    o->taper_end_percent = 1.0;
    o->taper_final_amplitude = 0.0;
    o->export_plot_csv = 0;
}

int hc_set_diagnostics_output_directory(hc_ctx* c, const char* dir) {
    HC_API_BEGIN(c)
    c->diagnostics_dir = dir ? dir : "";
    HC_API_END(c)
}

int hc_set_tapered_direct_options(hc_ctx* c, const hc_tapered_direct_options* o) {
    HC_API_BEGIN(c)
    require(o, HC_ERR_INVALID, "null options");
    require(o->smoothing == 0 || o->smoothing == 1, HC_ERR_INVALID, "smoothing must be 0 (sg) or 1 (moving_average)");
    c->taper      = *o;
    c->proc_ready = false;
    c->plan.valid = false;
    HC_API_END(c)
}

// ---- synthetic inputs generated in HBM --------------------------------------------------------
int hc_synth_fill(hc_ctx* c, unsigned long long seed, int S, double dt_rirf, int n_exc, double dt_exc) {
    HC_API_BEGIN(c)
    require(!c->finalized, HC_ERR_INVALID, "context already finalized");
    require(S > 0 && dt_rirf > 0, HC_ERR_INVALID, "bad synthetic sizes");
    if (!c->have_sim) {
        c->rho = 1000.0;
        c->g = 9.81;
        c->depth = std::numeric_limits<double>::infinity();
        c->have_sim = true;
    }
    c->S = S;
    c->tau.resize(S);
    for (int s = 0; s < S; ++s) c->tau[s] = s * dt_rirf;
    setup_panel_geometry(c);
    c->dK.alloc(hc::panel_doubles(c->ntiles, c->ngp));
    {
        KernelTimer kt(c->stream);
        hc::launch_synth_rirf(c->dK.p, c->ntiles, c->ngp, c->Dloc, c->D, S, 6 * c->b0, dt_rirf, seed, c->rho, c->stream);
        HC_HIP(hipGetLastError());
        c->init.synth_seconds += kt.stop();
        c->init.synth_bytes += 8.0 * static_cast<double>(c->dK.n);
    }
    // small per-body tables from the same counter-based stream, on the host
    auto mix = [](uint64_t x) {
        x += 0x9E3779B97F4A7C15ull;
        x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
        x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
        return x ^ (x >> 31);
    };
    auto u01 = [](uint64_t h) { return static_cast<double>(h >> 11) * (1.0 / 9007199254740992.0); };
    for (int b = 0; b < c->N; ++b) {
        auto& bd = c->bodies[b];
        const uint64_t hb = mix(seed ^ (0xB0D1ull << 40) ^ static_cast<uint64_t>(b));
        bd.disp_vol = 200.0 + 100.0 * u01(mix(hb + 1));
        for (int k = 0; k < 3; ++k) {
            bd.cg[k] = (k == 2 ? -2.0 : 20.0 * (b % 8) * (k == 0) + 20.0 * (b / 8) * (k == 1));
            bd.cb[k] = bd.cg[k] + (k == 2 ? 0.1 : 0.0);
        }
        for (int i = 0; i < 6; ++i)
            for (int j = 0; j < 6; ++j) {
                const double r = u01(mix(hb + 100 + 6 * std::min(i, j) + std::max(i, j)));
                bd.lin[6 * i + j] = (i == j ? 50.0 + 50.0 * r : 2.0 * (r - 0.5));
            }
        bd.have_props = bd.have_lin = bd.have_ainf = bd.have_rirf = true;
        if (is_local(c, b)) {
            bd.ainf.resize(static_cast<size_t>(6) * c->D);
            for (int i = 0; i < 6; ++i)
                for (int j = 0; j < c->D; ++j) {
                    const int gi = 6 * b + i;
                    const uint64_t h = mix(seed ^ (0xA1ull << 48) ^ (static_cast<uint64_t>(std::min(gi, j)) << 24) ^ static_cast<uint64_t>(std::max(gi, j)));
                    const double r = u01(h);
                    bd.ainf[static_cast<size_t>(i) * c->D + j] = c->rho * (gi == j ? 100.0 + 50.0 * r : (r - 0.5) * (j / 6 == b ? 5.0 : 0.5));
                }
            if (n_exc > 0) {
                bd.exirf_f.resize(static_cast<size_t>(6) * n_exc);
                for (int i = 0; i < 6; ++i) {
                    const uint64_t h = mix(seed ^ (0xE7ull << 48) ^ static_cast<uint64_t>(6 * b + i));
                    const double a = 1.0 + u01(mix(h + 1)), wd = 1.0 + 2.0 * u01(mix(h + 2)), om = 0.5 + 1.5 * u01(mix(h + 3));
                    for (int j = 0; j < n_exc; ++j) {
                        const double tt = (j - (n_exc - 1) * 0.5) * dt_exc;
                        bd.exirf_f[static_cast<size_t>(i) * n_exc + j] = c->rho * c->g * a * std::exp(-(tt * tt) / (wd * wd)) * std::cos(om * tt);
                    }
                }
            }
        }
        if (n_exc > 0) {
            bd.exirf_t.resize(n_exc);
            for (int j = 0; j < n_exc; ++j) bd.exirf_t[j] = (j - (n_exc - 1) * 0.5) * dt_exc;
            bd.have_exirf = true;
        }
    }
    HC_HIP(hipStreamSynchronize(c->stream));
    HC_API_END(c)
}

int hc_get_init_stats(const hc_ctx* c, hc_init_stats* out) {
    if (!c || !out) return HC_ERR_INVALID;
    *out = c->init;
    return HC_OK;
}

}  // extern "C"
