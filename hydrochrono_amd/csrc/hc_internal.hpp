// hc_internal.hpp -- what the translation units behind the C ABI share (hc_runtime.cpp: buffers, history ring, profiling, launch
// tiling, TaperedDirect preprocessing, direct-dispatch setup; hc_step.cpp: the per-step path; hc_setup.cpp: lifecycle, ingest,
// wave models; hc_query.cpp: introspection).  Not part of the public interface.
#pragma once
#include <dlfcn.h>
#include <fcntl.h>
#include <unistd.h>
#include <xmmintrin.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <limits>
#include <memory>
#include <string>
#include <vector>

#include "hc_context.hpp"
#include "hc_history.hpp"
#include "hc_host_math.hpp"

namespace hc {
void eta_synthesis_fft(const std::vector<double>& t, const std::vector<double>& amp, const std::vector<double>& omega,
                       const std::vector<double>& phase, double ramp, const double* d_t, double* d_eta, hipStream_t stream);

namespace detail {

using hc::Error;

extern thread_local std::string g_create_error;  // message of the last failed hc_create (hc_last_error(NULL))
extern const char* const kVersion;

#define HC_API_BEGIN_HOT(ctx)                                \
    if (!(ctx)) return HC_ERR_INVALID;                       \
    try {                                                    \
        HC_HIP(hipSetDevice((ctx)->device));

// every entry point but the per-step ones first waits for what the direct queue still runs (hc_direct.hpp): their HIP work is
// not ordered against it
#define HC_API_BEGIN(ctx)                                    \
    HC_API_BEGIN_HOT(ctx)                                    \
    quiesce_direct(ctx);

#define HC_API_END(ctx)                                      \
    }                                                        \
    catch (const Error& e) {                                 \
        (ctx)->err = e.what();                               \
        return e.status;                                     \
    }                                                        \
    catch (const std::out_of_range& e) {                     \
        (ctx)->err = e.what();                               \
        return HC_ERR_OUT_OF_RANGE;                          \
    }                                                        \
    catch (const std::exception& e) {                        \
        (ctx)->err = e.what();                               \
        return HC_ERR_RUNTIME;                               \
    }                                                        \
    return HC_OK;


// ---- hc_runtime.cpp ----
std::string device_local_cpus(int device);      // the kernel's list of CPUs local to the device's PCIe root ("" when unknown)
bool bind_calling_thread_to_device(int device); // restricts the calling thread to them (hc_bind_thread_to_device; the fan-out's workers)
int contexts_on_device(int device);             // contexts this process holds on a device (hc_create_sharded / hc_destroy keep count)
void count_context_on_device(int device, int delta);
void quiesce_direct(hc_ctx* c);  // waits for what the direct queue still runs (bounded; HC_ERR_DEVICE on a lost device)
void require(bool cond, int status, const char* msg);
void check_body(const hc_ctx* c, int body);
bool is_local(const hc_ctx* c, int body);
void ring_alloc(hc_ctx* c, int cap);
void ring_grow(hc_ctx* c, int need, int have);
int history_push(hc_ctx* c, double t);
void profile_account(hc_ctx* c, int kind, double sec, double waves_share);
void profile_drain(hc_ctx* c);
void profile_begin_step(hc_ctx* c);
hc::EventPair* ev_begin(hc_ctx* c, int kind, hipStream_t stream, double waves_share = 0.0);
void ev_end(hc::EventPair* ev, hipStream_t stream);
bool profiling_tool_attached();
int direct_tag(const hc_ctx* c, int kind);
int env_int(const char* name, int fallback);
// Environment switches.  The RELEASE library reads the operational ones only (env_int: HC_DIRECT, HC_ARM, HC_PASS_AHEAD,
// HC_PASS_AHEAD_GAP_US, HC_PASS_CONCURRENT, HC_DEVICE_SHARED, HC_MULTI_THREADS, HC_MULTI_SPIN_US; HC_STEP_TIMEOUT_S in hc_step.cpp --
// INTEGRATION.md lists them; tests/test_capi_exports.py checks the list against the strings of the built library).  Everything that
// exists for sweeps, A/B runs and fault injection -- tile counts, chunk lengths, kernel variants, the canary switch -- is read by the
// TUNING build only (-DHC_TUNING: libhydrochrono_amd_tuning.so + hc_kernels_tuning.co, which the tests that need a knob load);
// in the release build the macro is its default and the name does not exist.
#ifdef HC_TUNING
#define HC_TUNE_INT(name, fallback) ::hc::detail::env_int(name, fallback)
#else
#define HC_TUNE_INT(name, fallback) (fallback)
#endif
void setup_panel_geometry(hc_ctx* c);
hc::Panel rad_panel(const hc_ctx* c);
void choose_conv_config(hc_ctx* c);
void choose_exc_config(hc_ctx* c);
void alloc_partials(hc_ctx* c);
int far_chunk_gp(const hc_ctx* c);
int far_chunks_per_slice(const hc_ctx* c);
int default_pass_slices(const hc_ctx* c);
int default_pass_ahead(const hc_ctx* c);
bool pass_ahead_size_ok(const hc_ctx* c);
bool pass_ahead_possible(const hc_ctx* c);
void reset_schedule_state(hc_ctx* c);
void ensure_processed(hc_ctx* c);
void check_device_flag(hc_ctx* c);
void stage_state(hc_ctx* c, const double* pos, const double* rpy, const double* linvel, const double* angvel);
std::string library_dir();
bool direct_selftest_rewrites(hc_ctx* c, hc::DirectQueue* q, int lane, bool* abandon);
void setup_direct(hc_ctx* c);

// ---- hc_step.cpp ----
struct StepFlags {
    bool hs = true, rad = true, waves = true;
    bool scratch_out = false;  // term-only entry points: outputs go to scratch buffers, the last step's components stay
};
// The caller's state of a synchronous step, as handed to hc_step (host pointers, valid for the call), and the canary word of the step.
struct HostState {
    const double *pos, *rpy, *linvel, *angvel;
    double canary;
};
void enqueue_step(hc_ctx* c, double t, const double* d_state, double* d_user_out, hipStream_t stream, StepFlags f,
                  unsigned long long* host_tagged = nullptr, unsigned long long seq = 0, bool defer_tail = false, const HostState* host_state = nullptr);
void enqueue_tail(hc_ctx* c);

// ---- hc_step.cpp helpers the passes use ----
bool wave_window_ok(const hc_ctx* c, double t);
int live_samples(const hc_ctx* c, double t_query);
double* rows_P(hc_ctx* c, bool next);
double* rows_E(hc_ctx* c, bool next);

// ---- hc_pass.cpp: the look-ahead passes ----
struct StepViews {
    hc::Panel kex;
    hc::EtaTable ex;
};
StepViews make_views(const hc_ctx* c);
struct PassSetup {
    hc::BlockArgs b;
    bool exc_block = false;
    double rad_once = 0.0, exc_once = 0.0, bytes_steps = 0.0;
};
PassSetup make_pass(hc_ctx* c, bool with_exc, bool next_block);
void issue_pass_chunks(hc_ctx* c, const PassSetup& ps, int first, int last, bool with_items, hipStream_t stream, bool direct, int lane = 0);
void issue_pass_reduce(hc_ctx* c, const PassSetup& ps, double* P, double* E, hipStream_t stream, bool direct, int lane = 0);
void launch_pass(hc_ctx* c, hipStream_t stream, bool with_exc, bool direct = false);
void launch_mini_pass(hc_ctx* c, int i0, hipStream_t stream, bool direct, int next_kw = 0, int lane = 0);
void ahead_drop(hc_ctx* c);
void ahead_issue_slice(hc_ctx* c, hipStream_t stream, bool direct);
void ahead_begin(hc_ctx* c, hipStream_t stream, bool with_exc, bool direct);
bool ahead_expected(const hc_ctx* c, unsigned long long ended_serial);
bool ahead_adoptable(const hc_ctx* c, unsigned long long ended_serial);
bool pass_lane_ready(hc_ctx* c);
void pass_lane_drain(hc_ctx* c);

}  // namespace detail
}  // namespace hc
