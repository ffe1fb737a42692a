// hc_runtime.cpp -- host-side plumbing behind the C ABI (include/hydrochrono_amd.h): host-addressable device buffers, the history
// ring, profiling, launch tiling, TaperedDirect preprocessing, set-up of the direct AQL dispatch.  No per-step arithmetic here:
// that runs in hc_kernels.hip on the GPU; there is no CPU fallback.
#include "hc_internal.hpp"

#include <sched.h>

#include <cctype>

#include <atomic>

using namespace hc::detail;

// Fine-grained device allocation + a fault-free test of whether the CPU can address it: read(2) from /dev/zero INTO the
// buffer and write(2) FROM it fail with EFAULT instead of raising SIGSEGV when the range is not mapped for the host.
template <class T>
void hc::BarBuffer<T>::alloc(size_t count) {
    if (p) (void)hipFree(p);
    p       = nullptr;
    n       = 0;
    host_ok = false;
    if (count == 0) return;
    static const bool disabled = HC_TUNE_INT("HC_NO_BAR_STATE", 0) != 0;
    if (disabled) return;
    void* q = nullptr;
    if (hipExtMallocWithFlags(&q, count * sizeof(T), hipDeviceMallocFinegrained) != hipSuccess) {
        (void)hipGetLastError();
        return;
    }
    p = static_cast<T*>(q);
    n = count;
    const int fz = open("/dev/zero", O_RDONLY), fn = open("/dev/null", O_WRONLY);
    if (fz >= 0 && fn >= 0) {
        const size_t bytes = count * sizeof(T);
        host_ok = read(fz, p, bytes) == static_cast<ssize_t>(bytes) && write(fn, p, bytes) == static_cast<ssize_t>(bytes);
    }
    if (fz >= 0) close(fz);
    if (fn >= 0) close(fn);
}
template struct hc::BarBuffer<double>;

namespace hc {
namespace detail {

thread_local std::string g_create_error;

const char* const kVersion = "hydrochrono_amd 0.1 (gfx950)";

namespace {
std::atomic<int> g_contexts_on_device[64];
}
int contexts_on_device(int device) {
    // HC_DEVICE_SHARED=1: other PROCESSES hold contexts on the devices of this one too (one-process-per-rank runs that share a GPU, a
    // test bed) -- what the per-process counter cannot see; it counts as one more context everywhere
    static const int others = env_int("HC_DEVICE_SHARED", 0) != 0 ? 1 : 0;
    return ((device >= 0 && device < 64) ? g_contexts_on_device[device].load(std::memory_order_relaxed) : 0) + others;
}
void count_context_on_device(int device, int delta) {
    if (device >= 0 && device < 64) g_contexts_on_device[device].fetch_add(delta, std::memory_order_relaxed);
}

// ---- where the stepping thread should run (hc_device_local_cpus / hc_bind_thread_to_device) -------------------------------------
// /sys/bus/pci/devices/<domain:bus:device.function>/local_cpulist of the HIP device: the CPUs of the NUMA node its PCIe root hangs on.
std::string device_local_cpus(int device) {
    char bdf[64] = {0};
    if (hipDeviceGetPCIBusId(bdf, sizeof bdf, device) != hipSuccess) {
        (void)hipGetLastError();
        return "";
    }
    for (char* p = bdf; *p; ++p) *p = static_cast<char>(std::tolower(static_cast<unsigned char>(*p)));
    std::ifstream f(std::string("/sys/bus/pci/devices/") + bdf + "/local_cpulist");
    std::string line;
    if (!f || !std::getline(f, line)) return "";
    while (!line.empty() && (line.back() == '\n' || line.back() == ' ')) line.pop_back();
    return line;
}

bool bind_calling_thread_to_device(int device) {
    const std::string cpus = device_local_cpus(device);
    if (cpus.empty()) return false;
    cpu_set_t set;
    CPU_ZERO(&set);
    int n = 0;
    const char* p = cpus.c_str();
    while (*p) {  // "a-b,c,d-e"
        char* end = nullptr;
        const long a = std::strtol(p, &end, 10);
        if (end == p) break;
        long b = a;
        p = end;
        if (*p == '-') {
            b = std::strtol(p + 1, &end, 10);
            p = end;
        }
        for (long k = a; k <= b && k < CPU_SETSIZE; ++k) {
            if (k >= 0) { CPU_SET(static_cast<int>(k), &set); ++n; }
        }
        if (*p == ',') ++p;
    }
    // only CPUs this thread may use anyway (a container's cpuset): the intersection, and nothing if it is empty
    cpu_set_t now;
    CPU_ZERO(&now);
    if (sched_getaffinity(0, sizeof now, &now) == 0) {
        cpu_set_t both;
        CPU_AND(&both, &set, &now);
        if (CPU_COUNT(&both) == 0) return false;
        set = both;
    }
    return n > 0 && sched_setaffinity(0, sizeof set, &set) == 0;
}

// Bounded: a queue that does not drain within HC_STEP_TIMEOUT_S (or reports an error) is a lost device -> HC_ERR_DEVICE.
void quiesce_direct(hc_ctx* c) {
    static const double limit = [] { const char* e = std::getenv("HC_STEP_TIMEOUT_S"); const double v = e ? std::atof(e) : 0.0; return v > 0.0 ? v : 20.0; }();
    if (c->dq && c->dq->busy(2) && !c->dq->drain(limit, 2)) {  // the pass lane first: its work may wait for signals of lane 0
        // (lane 0 is still live, so those signals do arrive; a pass lane that does not drain is a lost device all the same)
        c->lost         = true;
        c->direct_ready = false;
        c->direct_why   = "the pass lane of the direct queue did not drain: " + c->dq->failure_text();
        throw Error(HC_ERR_DEVICE, c->direct_why);
    }
    if (c->dq && c->dq->busy()) {
        if (!c->dq->drain(limit)) {
            c->lost         = true;
            c->direct_ready = false;
            c->direct_why   = c->dq->failed(0) ? "the HSA queue of the direct dispatch reported an error: " + c->dq->failure_text()
                                               : std::string("the direct queue did not drain (timeout)");
            throw Error(HC_ERR_DEVICE, c->direct_why);
        }
    }
    if (c->path == 2) c->path = 1;
}

void require(bool cond, int status, const char* msg) {
    if (!cond) throw Error(status, msg);
}

void check_body(const hc_ctx* c, int body) {
    if (body < 0 || body >= c->N) throw Error(HC_ERR_OUT_OF_RANGE, "body index out of range");
}
bool is_local(const hc_ctx* c, int body) { return body >= c->b0 && body < c->b1; }

// ---- history ring -----------------------------------------------------------------------------
void ring_alloc(hc_ctx* c, int cap) {
    c->d_ring_t.alloc(cap);
    c->d_ring_v.alloc(static_cast<size_t>(cap) * c->D);
    c->d_ring_vT.alloc(static_cast<size_t>(cap + 2) * c->D);  // rows of Hcap + 2: entry [Hcap] mirrors slot 0 (hc_kernels.hpp)
    HC_HIP(hipMemsetAsync(c->d_ring_t.p, 0, cap * sizeof(double), c->stream));
    HC_HIP(hipMemsetAsync(c->d_ring_v.p, 0, static_cast<size_t>(cap) * c->D * sizeof(double), c->stream));
    HC_HIP(hipMemsetAsync(c->d_ring_vT.p, 0, static_cast<size_t>(cap + 2) * c->D * sizeof(double), c->stream));
    c->Hcap  = cap;
    c->HcapT = cap + 2;
    c->head = -1;
}

// Grow the ring so that `need` samples fit, keeping the `have` newest stored samples (k = 0..have-1) in order.
void ring_grow(hc_ctx* c, int need, int have) {
    quiesce_direct(c);               // the scatter / pass of the last step may still be reading the ring on the direct queue
    HC_HIP(hipDeviceSynchronize());  // rare; steps may have been enqueued on a caller's stream (hc_step_device)
    const int cap2 = std::max(2 * c->Hcap, need + 16);
    hc::DeviceBuffer<double> nt, nv;
    nt.alloc(cap2);
    nv.alloc(static_cast<size_t>(cap2) * c->D);
    HC_HIP(hipMemsetAsync(nt.p, 0, cap2 * sizeof(double), c->stream));
    HC_HIP(hipMemsetAsync(nv.p, 0, static_cast<size_t>(cap2) * c->D * sizeof(double), c->stream));
    // new layout: sample k -> slot (have-1-k); oldest at slot 0, newest at slot have-1
    for (int k = 0; k < have; ++k) {
        const int src = ((c->head - k) % c->Hcap + c->Hcap) % c->Hcap;
        const int dst = have - 1 - k;
        HC_HIP(hipMemcpyAsync(nt.p + dst, c->d_ring_t.p + src, sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        HC_HIP(hipMemcpyAsync(nv.p + static_cast<size_t>(dst) * c->D, c->d_ring_v.p + static_cast<size_t>(src) * c->D,
                              c->D * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    }
    c->d_ring_vT.alloc(static_cast<size_t>(cap2 + 2) * c->D);
    HC_HIP(hipMemsetAsync(c->d_ring_vT.p, 0, static_cast<size_t>(cap2 + 2) * c->D * sizeof(double), c->stream));
    hc::launch_ring_transpose(nv.p, cap2, cap2 + 2, c->D, c->d_ring_vT.p, c->stream);
    HC_HIP(hipGetLastError());
    HC_HIP(hipStreamSynchronize(c->stream));
    std::swap(c->d_ring_t.p, nt.p);
    std::swap(c->d_ring_t.n, nt.n);
    std::swap(c->d_ring_v.p, nv.p);
    std::swap(c->d_ring_v.n, nv.n);
    c->Hcap  = cap2;
    c->HcapT = cap2 + 2;
    c->head  = have - 1;
}

// Push the time of this step and prune like PruneHistory; returns H (samples incl. the current one).  The bookkeeping -- the
// reference's push / prune rule and what a step back in time does -- is hc_history.hpp (host only, unit-tested on CPU).
int history_push(hc_ctx* c, double t) {
    const hc::HistoryAdvance r = hc::history_advance(c->times, c->retired, c->head, c->Hcap, t, c->tau.empty() ? 0.0 : c->tau.back());
    if (r.status == hc::HistoryAdvance::kDuplicateTime)
        throw Error(HC_ERR_RUNTIME, "Tried to compute the radiation damping convolution twice within the same time step!");
    if (r.rewound) {
        // everything already enqueued (a scatter, a pass of the abandoned plan) runs before this step's kernels on the same queue,
        // so no ring slot is overwritten under a reader; the plan of the abandoned attempt is void
        // (the pass lane is not ordered against this step's kernels: a pass in the making -- void now -- must not read ring slots
        // while they are overwritten; a rewind is rare, so it is simply waited for)
        pass_lane_drain(c);
        c->ahead.active = false;
        c->plan.valid  = false;
        c->plan.misses = 0;
        c->rewinds++;
        c->prof.history_rewinds++;
    }
    if (r.grow) ring_grow(c, r.grow_need, r.grow_have);
    c->head = (c->head + 1) % c->Hcap;
    return r.H;
}

// ---- profiling --------------------------------------------------------------------------------
// Timed launches carry their own pair of HIP events, recorded on the stream the launch went to (a caller's stream in
// hc_step_device), so the drain waits on the events themselves, not on a particular stream.
void profile_account(hc_ctx* c, int kind, double sec, double waves_share) {
    switch (kind) {
        case hc::kEvConvPlain:  // radiation (+ irregular-wave excitation chunks) of a plain step
            c->prof.conv_kernel_seconds += sec;
            c->prof.conv_kernel_launches += 1;
            c->prof.radiation_seconds += sec * (1.0 - waves_share);
            c->prof.waves_seconds += sec * waves_share;
            break;
        case hc::kEvPass:  // look-ahead pass: radiation part of a block of steps (+ their excitation force)
            c->prof.block_kernel_seconds += sec;
            c->prof.block_kernel_launches += 1;
            c->prof.radiation_seconds += sec * (1.0 - waves_share);
            c->prof.waves_seconds += sec * waves_share;
            break;
        case hc::kEvStep:  // the step kernel: reduction, own-sample part, hydrostatics, regular / spectral wave term
            c->prof.step_kernel_seconds += sec;
            c->prof.step_kernel_launches += 1;
            c->prof.hydrostatics_seconds += sec;
            break;
        case hc::kEvScatter:
            c->prof.scatter_kernel_seconds += sec;
            c->prof.scatter_kernel_launches += 1;
            c->prof.radiation_seconds += sec;
            break;
        case hc::kEvMiniPass:  // short pass of the two-level form (one per sub-block of a wide system)
            c->prof.mini_pass_seconds += sec;
            c->prof.mini_pass_launches += 1;
            c->prof.radiation_seconds += sec;
            break;
        default:  // excitation-only convolution launch
            c->prof.waves_seconds += sec;
            break;
    }
}

void profile_drain(hc_ctx* c) {
    for (size_t i = 0; i < c->events_used; ++i) {
        hc::EventPair& ev = c->events[i];
        HC_HIP(hipEventSynchronize(ev.b));
        float ms = 0;
        HC_HIP(hipEventElapsedTime(&ms, ev.a, ev.b));
        profile_account(c, ev.kind, ms * 1e-3, ev.waves_share);
    }
    c->events_used = 0;
    // dispatches of the direct queue carry completion signals instead of events
    if (c->dq && c->dq->timed_pending() > 0)
        c->dq->collect([c](int kind, double sec, double share) { profile_account(c, kind, sec, share); });
}

constexpr size_t kEventPoolMax = 4096;

// Called at the top of a step, before anything is enqueued or the history is touched: decides whether this step's launches
// are timed (every stride-th step) and makes room in the event pool (draining synchronises, so it must not happen between
// the history push and the launches).
void profile_begin_step(hc_ctx* c) {
    c->sample_this_step = false;
    if (!c->profiling) return;
    if (c->events_used + 8 > kEventPoolMax || (c->dq && c->dq->timed_pending() + 8 > kEventPoolMax)) profile_drain(c);
    c->sample_this_step = (c->profile_counter++ % c->profile_stride) == 0;
}

// Event pair around one launch, or null.  The look-ahead pass (one per block) is timed whatever the stride.
hc::EventPair* ev_begin(hc_ctx* c, int kind, hipStream_t stream, double waves_share) {
    if (!c->profiling || !(c->sample_this_step || kind == hc::kEvPass)) return nullptr;
    if (c->events_used == c->events.size()) {
        if (c->events.size() >= kEventPoolMax) return nullptr;
        hc::EventPair ev;
        HC_HIP(hipEventCreate(&ev.a));
        HC_HIP(hipEventCreate(&ev.b));
        c->events.push_back(ev);
    }
    hc::EventPair* ev = &c->events[c->events_used++];
    ev->kind        = kind;
    ev->waves_share = waves_share;
    HC_HIP(hipEventRecord(ev->a, stream));
    return ev;
}
// A profiling tool that intercepts the HSA queues (rocprofv3, roctracer: they arrive through these variables) replaces the
// completion signals of our packets with its own, and hsa_amd_profiling_get_dispatch_time on ours then returns nothing useful
// (1.3 us for a 190 us pass).  The tool still sees every dispatch; only the library's own timings need the HIP path then.
bool profiling_tool_attached() {
    static const bool attached = [] {
        if (std::getenv("ROCP_TOOL_LIBRARIES") || std::getenv("HSA_TOOLS_LIB")) return true;
        const char* pre = std::getenv("LD_PRELOAD");
        return pre && (std::strstr(pre, "rocprofiler") || std::strstr(pre, "roctracer"));
    }();
    return attached;
}

// direct dispatches: the tag to time a launch with (-1: not timed), same sampling rule as ev_begin
int direct_tag(const hc_ctx* c, int kind) { return (c->profiling && (c->sample_this_step || kind == hc::kEvPass)) ? kind : -1; }

void ev_end(hc::EventPair* ev, hipStream_t stream) {
    if (ev) HC_HIP(hipEventRecord(ev->b, stream));
}

// ---- panel geometry and launch tiling ---------------------------------------------------------
int env_int(const char* name, int fallback) {
    const char* e = std::getenv(name);  // HC_* variables are tuning knobs for experiments only
    return e ? std::atoi(e) : fallback;
}

void setup_panel_geometry(hc_ctx* c) {
    c->ntiles = (c->Dloc + 15) / 16;
    c->Dpad   = c->ntiles * 16;
    c->ngp    = static_cast<int>((static_cast<long long>(c->S) * c->D + 7) / 8);
    c->mt     = (c->ntiles % 4 == 0) ? 4 : ((c->ntiles % 2 == 0) ? 2 : 1);
    const int want = HC_TUNE_INT("HC_CONV_MT", 0);
    if ((want == 1 || want == 2 || want == 4) && c->ntiles % want == 0) c->mt = want;
    c->ngroups = c->ntiles / c->mt;
    // Look-ahead pass: as many row tiles per workgroup as the tile count allows -- every workgroup of a chunk forms the same
    // B operands (ring gathers + interpolation), so fewer, taller workgroups repeat less of that work; it is what bounds
    // the pass once K streams at the HBM rate (C3: 415 / 277 us with 4 / 6 tiles at depth 32 before the per-DoF ring).
    // The design value comes from the UNSHARDED tile count (the chunk length below must not depend on the rows owned).
    auto pick = [](int tiles, int limit) {
        for (int m : {12, 6, 4, 2, 1})
            if (m <= limit && tiles % m == 0) return m;
        return 1;
    };
    const int tiles_full  = (c->D + 15) / 16;
    const int limit       = HC_TUNE_INT("HC_BLOCK_MT", 6);
    c->mt_block_design    = pick(tiles_full, limit);
    c->mt_block           = pick(c->ntiles, c->mt_block_design);
    // Short passes of the two-level form stream a few tens of IRF samples only: with the pass's tall workgroups (6 tiles x half a
    // sample = 1.2 MB each at D = 3072) there are fewer workgroups than CUs and each streams for ~50 us; fewer tiles per workgroup
    // give a multiple of the workgroups, each done sooner (the B-operand work they repeat is small here).
    // Tiles per workgroup = the largest that still leaves about two workgroups per CU (a short pass has roughly 48 chunks of half a
    // sample): 2 for a C4/8 shard (24 tiles: 76 -> 68 us per short pass), 6 for C4 on one GPU (192 tiles: 377 us; 422 with 2).
    {
        const int forced = HC_TUNE_INT("HC_MINI_MT", 0);
        int m_pick = 1;
        for (int m : {1, 2, 4, 6})
            if (m <= c->mt_block && c->ntiles % m == 0 && 48LL * (c->ntiles / m) >= 2LL * c->num_cus) m_pick = m;
        c->mt_mini = forced > 0 ? pick(c->ntiles, std::min(c->mt_block, forced)) : m_pick;
        // ... and of their narrow form (16 step columns, two to three waves per SIMD): taller workgroups pay, as long as a short pass still
        // has a workgroup for every CU -- 4 for a C4/8 shard (55 -> 51 us per short pass; 67 us in the wide form), 6 for C4 on one GPU
        const int forced_n = HC_TUNE_INT("HC_NARROW_MT", 0);
        int n_pick = 1;
        for (int m : {1, 2, 4, 6})
            if (m <= c->mt_block && c->ntiles % m == 0 && 48LL * (c->ntiles / m) >= 1LL * c->num_cus) n_pick = m;
        c->mt_narrow = forced_n > 0 ? pick(c->ntiles, std::min(c->mt_block, forced_n)) : std::max(n_pick, c->mt_mini);
    }
}

hc::Panel rad_panel(const hc_ctx* c) {
    hc::Panel p;
    p.base   = (c->conv_mode == 1) ? c->dKproc.p : c->dK.p;
    p.ntiles = c->ntiles;
    p.ngp    = c->ngp;
    return p;
}

void choose_conv_config(hc_ctx* c) {
    // plain per-step kernel: 8 workgroups per CU (one resident set; 184.6 us against 187.2 us with twice as many, and
    // finalize_kernel has half as many partials to add); a chunk is a whole number of 8-column groups.  Like the pass
    // below, the chunk length is a function of the column count only (row groups of the UNSHARDED system), so row-sharded
    // contexts add their partial sums in the same order as the unsharded one (bitwise equal results).
    const int target_wgs       = std::max(1, HC_TUNE_INT("HC_CONV_TARGET_WGS", 8 * c->num_cus));
    const long long rows_full  = std::max<long long>(1, (((c->D + 15) / 16) + 3) / 4);
    long long nch              = std::max<long long>({1, target_wgs / rows_full, c->num_cus / 2});
    long long gps        = (c->ngp + nch - 1) / nch;
    gps                  = std::max<long long>(16, ((gps + 3) / 4) * 4);  // every wave of the workgroup gets work
    gps                  = std::min<long long>(gps, 512);                 // the chunk's right-hand side is staged in LDS (<= 32 KB)
    c->chunk_gp          = static_cast<int>(gps);
    c->nchunks_rad       = static_cast<int>((c->ngp + gps - 1) / gps);
    // Look-ahead pass.  Its workgroups live long (tens of microseconds) and two fit on a CU, so the launch runs in "rounds" of
    // 2 * CUs workgroups and is fastest when row groups x chunks fills whole rounds: C3 has 4 row groups (6 tiles each), so
    // CUs / 2 = 128 chunks make exactly one round (207 us per pass against 221 us at three rounds and 249 us at one and a
    // half).  The chunk length depends on the column count only -- never on how many rows this context owns -- so that
    // row-sharded contexts add their partial sums in the same order as the unsharded one (bitwise equal results).
    long long bgps;
    const int forced = HC_TUNE_INT("HC_BLOCK_CHUNK_GP", 0);
    if (forced > 0) {
        bgps = std::max(8, forced);
        bgps = std::max<long long>(bgps, (c->ngp + 255) / 256);
    } else {
        // row groups of the UNSHARDED system (6 tiles each): few of them (small systems) need more chunks to fill a round.
        // Two workgroups fit on a CU at depth 16, one at depth 32 (twice the accumulators).
        const int mt_design         = c->lookahead > 32 ? c->mt_block64 : c->mt_block_design;
        const long long groups_full = std::max<long long>(1, ((c->D + 15) / 16 + mt_design - 1) / mt_design);
        const long long slots       = ((c->lookahead > 16 || c->mt_block_design > 6) ? 1LL : 2LL) * c->num_cus;
        const long long nch_target  = std::max<long long>(slots / 4, (slots + groups_full - 1) / groups_full);
        bgps                       = (c->ngp + nch_target - 1) / nch_target;
        bgps                       = std::min<long long>(bgps, std::max<long long>(16, (16LL * c->D) / 8));  // <= 16 IRF samples per chunk
    }
    const long long cap = std::max<long long>(8, (64LL * c->D) / 8);   // bracket table [samples][16] must fit in LDS
    bgps                = std::min(bgps, cap);
    bgps                = std::max<long long>(16, ((bgps + 15) / 16) * 16);  // whole 16-group sub-tiles
    c->chunk_gp_block   = static_cast<int>(bgps);
    c->nchunks_block    = static_cast<int>((c->ngp + bgps - 1) / bgps);
}

// Chunk length of a pass that is issued in slices (pass schedule "one block ahead").  The pass of a block is ONE round of long-lived
// workgroups -- one per CU at depth 32, each streaming its chunk for the whole duration -- so a slice must bring a full round of its
// own, and the round is that of the PASS LANE, whose queue leaves pass_free_cus compute units of every XCD to the step kernels
// (hc_step.cpp: pass_lane_ready): with the usual four row groups a slice has (CUs - 8 * free) / 4 chunks, in whole octets (56 on
// an MI355X with 4 free CUs per XCD).  A function of the column count, the slice count and the device only, like every chunk length.
// Default pass schedule (hc_set_pass_schedule): ADAPTIVE (2) at every size -- per block, from the gaps the caller leaves between
// its synchronous steps (hc_pass.cpp: schedule_ahead_for_next_block).  A caller that steps back to back has nothing to hide the pass
// behind and is best served by the pass at block start (C3: 18.4 against 19-20 us per step; the slices also run at 0.62 of the HBM
// peak against 0.78 unsliced); a caller that stays away between two force evaluations -- every Chrono loop -- never waits for a pass
// when it runs one block ahead (C3, 30 us of host work: 18.9 -> 14.2 us mean, worst step 183 -> 22 us; a C4/8 rank, 300 us:
// 60 -> 21 us).  HC_PASS_AHEAD=0/1 pins new contexts to one schedule.
int default_pass_ahead(const hc_ctx* c) {
    (void)c;
    const int forced = env_int("HC_PASS_AHEAD", -1);
    if (forced >= 0) return forced != 0 ? 1 : 0;
    return 2;
}

// The adaptive schedule considers "one block ahead" only where a pass is long enough to matter: 256 MB of (unsharded) K and more,
// i.e. passes of 40 us and up.  Below that (the reference's own one- to three-body demos: 0.3 - 2.6 MB) the pass takes a few
// microseconds, any gap that would select the schedule hides it anyway, and the schedule's extra launches would sit between the
// steps of a caller that comes back within microseconds (one body: 8.9 -> 10.1 us per step).  A function of D and S only: the
// row shards of one array answer alike.
bool pass_ahead_size_ok(const hc_ctx* c) {
    const double floor_bytes = 1e6 * HC_TUNE_INT("HC_PASS_AHEAD_MIN_MB", 256);  // (tests run the rule on small systems with 0; read per call: once per block)
    return 8.0 * static_cast<double>(c->D) * c->D * c->S >= floor_bytes;
}

// may a pass one block ahead ever run under the selected schedule (buffers, pass lane)?
bool pass_ahead_possible(const hc_ctx* c) { return c->pass_ahead == 1 || (c->pass_ahead == 2 && pass_ahead_size_ok(c)); }

// Default slice count: slices of roughly 320 MB of the UNSHARDED K (a slice then takes 50-200 us on one GPU or on a row shard), between
// 2 and 8 -- 4 at C3 (each launch has a fixed cost of about 11 us there: 8 slices make the pass 283 us instead of 192, 4 make it 234),
// 8 for C4.  A function of the system's size only, never of the rows a context owns.
int default_pass_slices(const hc_ctx* c) {
    const int forced = HC_TUNE_INT("HC_PASS_SLICES", 0);
    if (forced > 0) return std::min(forced, hc::kDepthDefault - 1);
    const double bytes = 8.0 * static_cast<double>(c->D) * c->D * c->S;
    return std::max(2, std::min(8, static_cast<int>(std::ceil(bytes / 320e6))));
}

int far_chunks_per_slice(const hc_ctx* c) {
    const int usable = c->num_cus - 8 * c->pass_free_cus;  // (also where the pass lane is not in use: one arithmetic for every way the slices are issued)
    return std::max(8, ((usable / 4) / 8) * 8);
}

int far_chunk_gp(const hc_ctx* c) {
    const long long nch = static_cast<long long>(far_chunks_per_slice(c)) * std::max(1, c->pass_slices);
    long long g         = (c->ngp + nch - 1) / nch;
    g                   = std::max<long long>(16, ((g + 15) / 16) * 16);
    return static_cast<int>(std::min<long long>(g, std::max(16, c->chunk_gp_block)));
}

void choose_exc_config(hc_ctx* c) {
    if (c->wave_kind != hc::kWaveIrregular || c->L == 0) {
        c->nchunks_ex  = 0;
        c->chunk_gp_ex = 64;
        c->nchunks_ex_block = 0;
        return;
    }
    c->chunk_gp_ex = std::max(4, HC_TUNE_INT("HC_EXC_CHUNK_GP", 8));  // short chunks: the excitation side is latency-bound
    c->nchunks_ex  = (c->ngp_ex + c->chunk_gp_ex - 1) / c->chunk_gp_ex;
    c->chunk_gp_ex_block = 16;  // one 16-group sub-tile of the look-ahead kernel per work item (the items are dealt one per workgroup)
    c->nchunks_ex_block  = (c->ngp_ex + c->chunk_gp_ex_block - 1) / c->chunk_gp_ex_block;
}

void alloc_partials(hc_ctx* c) {
    const size_t n = static_cast<size_t>(c->nchunks_rad + c->nchunks_ex) * c->Dpad;
    if (c->d_partials.n < n) c->d_partials.alloc(n);
    // (the short passes of the two-level form use the same buffer: IRF samples s < kScatterSamples -- the planner refuses blocks whose
    // in-block brackets reach further -- in chunks of at least half a sample)
    // (pass schedule "one block ahead": its short passes towards the next block reach twice as far, and the pass in the making keeps
    // a buffer of its own while the short passes of the current block use this one)
    const bool ahead_bufs = pass_ahead_possible(c);
    const int mini_chunks = (ahead_bufs ? 4 : 2) * hc::kScatterSamples + 2;
    const size_t nb = static_cast<size_t>(std::max(c->nchunks_block + c->nchunks_ex_block, mini_chunks)) * hc::kLookahead * c->Dpad;
    if (c->d_partials_block.n < nb) c->d_partials_block.alloc(nb);
    const size_t nfar = static_cast<size_t>((c->ngp + far_chunk_gp(c) - 1) / far_chunk_gp(c) + c->nchunks_ex_block) * hc::kLookahead * c->Dpad;
    if (ahead_bufs && c->d_partials_far.n < nfar) {
        // (a pass one block ahead in the making has left the chunk partials of its slices so far in the OLD buffer -- a wave model with more
        // excitation chunks attached in the middle of a block, profiles/fuzz_parity.py seed 2037 --: it is abandoned, the next block runs
        // its own pass)
        c->d_partials_far.alloc(nfar);
        c->ahead.active = false;
    }
    const size_t nnext = static_cast<size_t>(mini_chunks) * hc::kLookahead * c->Dpad;
    if (ahead_bufs && c->d_partials_next.n < nnext) c->d_partials_next.alloc(nnext);
    // two blocks of rows each: the current block's and (pass schedule "one block ahead") the next one's
    const size_t npe = static_cast<size_t>(2 * hc::kLookahead) * c->Dpad;
    if (c->d_P.n < npe) c->d_P.alloc(npe);
    if (c->d_E.n < npe) c->d_E.alloc(npe);
    const size_t ny = static_cast<size_t>(hc::kLookahead + 1) * hc::kTermMax * c->Dpad;
    if (c->d_Y.n < ny) c->d_Y.alloc(ny);
    if (c->d_near_partials.n < static_cast<size_t>(16) * c->Dpad) c->d_near_partials.alloc(static_cast<size_t>(16) * c->Dpad);
    if (c->d_tile_counter.n < static_cast<size_t>(c->ntiles)) {  // arrival counters of wide_step_kernel, zero between launches
        c->d_tile_counter.alloc(static_cast<size_t>(c->ntiles));
        HC_HIP(hipMemset(c->d_tile_counter.p, 0, static_cast<size_t>(c->ntiles) * sizeof(int)));
    }
}

// ---- TaperedDirect ----------------------------------------------------------------------------
// The diagnostics files of EnsureProcessedRIRF (src/hydro_forces.cpp:509-531): per body one rirf_body<b>_summary.csv (b 0-based)
// with the representative channel row 0 / column 0 before and after the processing, rows s < effective_steps, numbers in the
// default ostream format like the reference's `ofs << s << "," << t << "," << before << "," << after`.  Errors are ignored (:529).
void export_taper_csv(hc_ctx* c, int effective) {
    try {
        const int n = std::max(0, std::min(effective, c->S));
        hc::DeviceBuffer<double> d_before, d_after;
        d_before.alloc(std::max(1, n));
        d_after.alloc(std::max(1, n));
        std::vector<double> before(n), after(n);
        hc::Panel raw;
        raw.base   = c->dK.p;
        raw.ntiles = c->ntiles;
        raw.ngp    = c->ngp;
        hc::Panel proc = raw;
        proc.base      = c->dKproc.p;
        for (int bl = 0; bl < c->nloc; ++bl) {
            if (n > 0) {
                hc::launch_extract_series(raw, 6 * bl, 0, c->D, n, d_before.p, c->stream);
                hc::launch_extract_series(proc, 6 * bl, 0, c->D, n, d_after.p, c->stream);
                HC_HIP(hipMemcpyAsync(before.data(), d_before.p, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
                HC_HIP(hipMemcpyAsync(after.data(), d_after.p, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
                HC_HIP(hipStreamSynchronize(c->stream));
            }
            const std::string base = "rirf_body" + std::to_string(c->b0 + bl) + "_summary.csv";
            const std::string path = c->diagnostics_dir.empty() ? base : (c->diagnostics_dir + "/" + base);
            std::ofstream ofs(path);
            ofs << "step,time,k_before,k_after\n";
            for (int s_ = 0; s_ < n; ++s_) ofs << s_ << "," << c->tau[s_] << "," << before[s_] << "," << after[s_] << "\n";
        }
    } catch (...) {
        (void)hipGetLastError();  // "ignore export errors"
    }
}

void ensure_processed(hc_ctx* c) {
    if (c->conv_mode != 1 || c->proc_ready) return;
    const int steps = c->S;
    int effective   = steps;
    if (c->taper.rirf_end_time > 0.0) {
        require(c->S >= 2, HC_ERR_INVALID, "TaperedDirect truncation needs at least two IRF samples");
        const double dt = c->tau[1] - c->tau[0];
        const int end   = static_cast<int>(std::floor(c->taper.rirf_end_time / dt));
        effective       = std::min(end, steps);
    }
    int tc_index = static_cast<int>(std::floor(c->taper.taper_start_percent * static_cast<double>(effective)));
    int tc_end   = static_cast<int>(std::floor(c->taper.taper_end_percent * static_cast<double>(effective)));
    tc_index     = std::max(0, std::min(tc_index, effective));
    tc_end       = std::max(tc_index, std::min(tc_end, effective));
    HC_HIP(hipDeviceSynchronize());  // once per option change; orders against steps on a caller's stream
    if (c->dKproc.n != c->dK.n) {
        c->dKproc.alloc(c->dK.n);
        HC_HIP(hipMemsetAsync(c->dKproc.p, 0, c->dKproc.n * sizeof(double), c->stream));
    }
    hc::TaperArgs a{};
    a.Kraw.base       = c->dK.p;
    a.Kraw.ntiles     = c->ntiles;
    a.Kraw.ngp        = c->ngp;
    a.Kproc           = c->dKproc.p;
    a.Dloc            = c->Dloc;
    a.D               = c->D;
    a.S               = c->S;
    a.effective_steps = effective;
    a.smoothing       = c->taper.smoothing;
    a.window          = std::max(3, c->taper.window_length);
    a.tc_index        = tc_index;
    a.tc_end          = tc_end;
    a.final_amplitude = c->taper.taper_final_amplitude;
    {
        hipEvent_t e0 = nullptr, e1 = nullptr;
        const bool timed = hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess && hipEventRecord(e0, c->stream) == hipSuccess;
        hc::launch_taper(a, c->stream);
        HC_HIP(hipGetLastError());
        float ms = 0.0f;
        if (timed && hipEventRecord(e1, c->stream) == hipSuccess && hipEventSynchronize(e1) == hipSuccess) (void)hipEventElapsedTime(&ms, e0, e1);
        (void)hipGetLastError();
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
        c->init.taper_seconds += 1e-3 * static_cast<double>(ms);  // (hc_init_stats)
        c->init.taper_bytes += 2.0 * 8.0 * static_cast<double>(c->dK.n);
    }
    HC_HIP(hipStreamSynchronize(c->stream));
    c->proc_ready = true;
    c->plan.valid = false;
    if (c->taper.export_plot_csv) export_taper_csv(c, effective);
}

void check_device_flag(hc_ctx* c) {
    HC_HIP(hipMemcpyAsync(c->h_err.p, c->d_err.p, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HC_HIP(hipStreamSynchronize(c->stream));
    if (*c->h_err.p != 0) {
        const int code = *c->h_err.p;
        HC_HIP(hipMemsetAsync(c->d_err.p, 0, sizeof(int), c->stream));
        if (code == 1) throw Error(HC_ERR_RUNTIME, "Radiation convolution: interpolation error; query_time not bracketed by history.");
        throw Error(HC_ERR_RUNTIME, "Excitation convolution: tau value not bracketed by the free-surface table");
    }
}

void stage_state(hc_ctx* c, const double* pos, const double* rpy, const double* linvel, const double* angvel) {
    HC_HIP(hipStreamSynchronize(c->stream));  // kernels of the last hc_step may still be reading the pinned state (zero-copy)
    const int n3 = 3 * c->N;
    double* h    = c->h_state.p;
    const double* src[4] = {pos, rpy, linvel, angvel};
    for (int k = 0; k < 4; ++k) {
        if (src[k]) std::memcpy(h + k * n3, src[k], n3 * sizeof(double));
        else std::memset(h + k * n3, 0, n3 * sizeof(double));
    }
    HC_HIP(hipMemcpyAsync(c->d_state.p, h, 4 * n3 * sizeof(double), hipMemcpyHostToDevice, c->stream));
}

std::string library_dir() {
    Dl_info info;
    std::string dir = ".";
    if (dladdr(reinterpret_cast<void*>(&hc_version), &info) && info.dli_fname) {
        std::string full(info.dli_fname);
        const size_t slash = full.find_last_of('/');
        if (slash != std::string::npos) dir = full.substr(0, slash);
    }
    return dir;
}

// What the host RE-writes through the BAR must be what the next dispatch reads.  The dispatches carry agent-scope acquire fences
// only, so this rests on the GPU not keeping stale copies of fine-grained device memory across kernels -- for the state buffer
// (two halves, rewritten every second step) and for the kernarg ring (64 slots per lane, reused every 64 dispatches).  Checked
// on the given lane: 2 x 64 + 3 one-row added-mass products whose inputs (w, R_in: the SAME two BAR words every time) and
// whose argument slot change from dispatch to dispatch; every result must be the one of the values written last.  A stale read
// fails the test (reason in c->direct_why) and the caller keeps using HIP launches.  *abandon: a dispatch never completed.
bool direct_selftest_rewrites(hc_ctx* c, hc::DirectQueue* q, int lane, bool* abandon) {
    *abandon = false;
    const double one = 1.0;
    HC_HIP(hipMemcpy(c->d_selftest.p, &one, sizeof one, hipMemcpyHostToDevice));  // the 1 x 1 "matrix"
    volatile unsigned long long* tag = c->h_tag_selftest.p;
    tag[0] = tag[1] = 0;
    for (int i = 0; i < 2 * 64 + 3; ++i) {
        const double wv = 3.0 + i, rv = 0.25 * (i + 1) + lane, cv = 0.5 + 0.125 * (i % 7);
        c->bar_selftest.p[0] = wv;
        c->bar_selftest.p[1] = rv;
        _mm_sfence();
        const unsigned long long sq = 0xABC000ull + static_cast<unsigned long long>(lane) * 1000 + i;
        hc::AddedMassArgs a{c->d_selftest.p, 1, 1, c->bar_selftest.p, c->bar_selftest.p + 1, cv, c->h_tag_selftest.dp, sq, nullptr, nullptr};
        q->dispatch(c->dk_added_mass, 1, 256, 0, &a, sizeof a, -1, 0.0, lane);
        const auto t0 = std::chrono::steady_clock::now();
        bool arrived  = false;
        for (unsigned long long spins = 0;; ++spins) {
            if (tag[1] == sq) { arrived = true; break; }
            __builtin_ia32_pause();
            if ((spins & 0xFFF) == 0xFFF && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 2.0) break;
        }
        if (!arrived) {
            // either the dispatch hangs or it ran with the arguments of the slot's previous use (an old sequence number)
            if (q->drain(2.0, lane)) {
                c->direct_why = "self-test of the direct dispatch: a re-used kernel-argument slot was read stale";
            } else {
                c->direct_why = "self-test of the direct dispatch timed out";
                *abandon      = true;
            }
            return false;
        }
        const unsigned long long bits = tag[0];
        double got;
        std::memcpy(&got, &bits, sizeof got);
        if (got != rv + cv * wv) {
            c->direct_why = "self-test of the direct dispatch: memory re-written through the PCIe BAR was read stale";
            (void)q->drain(2.0, lane);
            return false;
        }
    }
    if (!q->drain(2.0, lane)) {
        c->direct_why = "self-test of the direct dispatch timed out";
        *abandon      = true;
        return false;
    }
    return true;
}

// Direct AQL dispatch for the synchronous step path (hc_direct.hpp).  Optional: when anything it needs is missing -- the code
// object next to the library, a host-addressable BAR, one of the kernels of this configuration -- the HIP launches stay in use
// and hc_last_error-style diagnostics keep the reason (HC_DEBUG_PLAN prints it).  Still the GPU path either way.
#ifdef HC_TUNING
static const char* const kKernelObjectName = "/hc_kernels_tuning.co";
#else
static const char* const kKernelObjectName = "/hc_kernels.co";
#endif
// The packet processor fetches AQL packets from the queue's ring; by default the HSA runtime puts that ring into HOST memory (a PCIe
// read in front of every dispatch), with HSA_ALLOCATE_QUEUE_DEV_MEM=1 into device memory: 1.0-1.8 us less from doorbell to kernel start,
// i.e. per synchronous hc_step and per hc_added_mass_mv (profiles/r06/queue_dev_mem_ab.txt, host_path_c.txt: 7.11 -> 6.15 us at one
// body, 9.97 -> 8.85 us at 64).  OPT-IN (HC_QUEUE_DEV_MEM=1), not the default: the runtime's variable moves the rings of EVERY queue of the
// process, the HIP runtime's included, and a doorbell can reach the packet processor before a packet stored through the BAR has left
// the HDP for VRAM.  This library orders its own packets (DirectQueue: HDP write-back in front of every doorbell, hc_direct.cpp); the HIP
// runtime's dispatch path does not, and round 6's long differential runs lost 2 of ~2 600 cases with the variable set -- a HIP queue
// aborting with "invalid code object", a memory fault at a garbage address: packets read half new, half old (EXPERIMENTS.md).  The
// runtime reads the variable once, when it is initialised, so a host that opts in gets it set when the library is LOADED -- before
// main() of a program linked against it, before the first HIP call of an interpreter that imports it first.
__attribute__((constructor)) static void request_device_memory_queue_rings() {
    if (env_int("HC_QUEUE_DEV_MEM", 0) == 0) return;
    (void)setenv("HSA_ALLOCATE_QUEUE_DEV_MEM", "1", 0);
}

void setup_direct(hc_ctx* c) {
    c->direct_ready = false;
    if (env_int("HC_DIRECT", 1) == 0) { c->direct_why = "disabled by HC_DIRECT=0"; return; }
    if (HC_TUNE_INT("HC_BLOCK_V32", 0) != 0) { c->direct_why = "HC_BLOCK_V32 selects a tuning variant of the pass"; return; }
    if (!c->bar_state.host_ok || !c->bar_am.host_ok || !c->bar_selftest.host_ok) { c->direct_why = "the device's memory is not host-addressable"; return; }
    std::unique_ptr<hc::DirectQueue> q(new hc::DirectQueue);
    std::string why;
    if (!q->init(c->device, library_dir() + kKernelObjectName, &why)) { c->direct_why = why; return; }
    c->dk_finalize = q->find("finalize_kernelILi4ELb0EEEv");
    c->dk_finalize_slot = q->find("finalize_kernelILi4ELb1EEEv");  // optional: the step kernel that finds the body state behind its arguments
    c->dk_scatter  = q->find("scatter_kernelE");
    c->dk_near     = q->find("near_split_kernelE");
    c->dk_wide     = q->find("wide_step_kernelE");  // optional: without it a wide step is near_split_kernel + finalize_kernel
    if (c->dk_wide.kernarg != sizeof(hc::WideStepArgs) || c->dk_wide.priv != 0) c->dk_wide = hc::DirectKernel{};
    if (c->dk_finalize_slot.kernarg != sizeof(hc::FinalizeArgs) || c->dk_finalize_slot.priv != 0) c->dk_finalize_slot = hc::DirectKernel{};
    c->slot_state = HC_TUNE_INT("HC_SLOT_STATE", 1) != 0 && c->dk_finalize_slot.ok() && c->N <= hc::kSlotStateMaxBodies;
#ifdef HC_TUNING
    c->dk_finalize_pre = q->find("finalize_pre_kernelILi4EEEv");  // (the kernel-argument-preload experiment of round 6)
    if (c->dk_finalize_pre.kernarg != sizeof(hc::FinalizePreArgs) || c->dk_finalize_pre.priv != 0) c->dk_finalize_pre = hc::DirectKernel{};
    c->step_preload = c->slot_state && c->dk_finalize_pre.ok() && HC_TUNE_INT("HC_STEP_PRELOAD", 0) != 0;
#endif
    c->dk_step_hot[0] = q->find("step_hot_kernelILi1EEEv");  // optional: without them the general step kernel runs every step
    c->dk_step_hot[1] = q->find("step_hot_kernelILi2EEEv");
    for (auto& k : c->dk_step_hot)
        if (k.kernarg != sizeof(hc::StepHotArgs) || k.priv != 0) k = hc::DirectKernel{};
    c->step_hot = c->slot_state && c->dk_step_hot[0].ok() && c->dk_step_hot[1].ok() && HC_TUNE_INT("HC_STEP_HOT", 1) != 0;
    c->step_halves = HC_TUNE_INT("HC_STEP_HALVES", 1) == 2 ? 2 : 1;
    c->dk_reduce   = q->find("reduce_block_kernelE");
    c->dk_added_mass = q->find("added_mass_mv_tagged_kernelE");  // optional: hc_added_mass_mv falls back to a HIP launch
    {   // the plain per-step convolution of this context's tiling; optional: without it plain steps go through HIP launches
        hc::StepArgs a{};
        a.ngroups = 1;
        const hc::StepLaunch l = hc::step_launch_config(a, c->mt);
        char frag[64];
        std::snprintf(frag, sizeof frag, "conv_step_kernelILi%dELi%dEEEv", l.MT, l.U);
        c->dk_step = q->find(frag);
        if (c->dk_step.kernarg != sizeof(hc::StepArgs) || c->dk_step.priv != 0) c->dk_step = hc::DirectKernel{};
    }
    for (int depth : {16, 32}) {
        hc::BlockArgs a{}, b{};
        a.depth = depth;
        a.ngroups = 1;
        const hc::BlockLaunch l = hc::block_launch_config(a, c->mt_block, &b);
        char frag[96];
        std::snprintf(frag, sizeof frag, "conv_block_kernelILi%dELi%dELi%dELi%dEEEv", l.MT, l.R, l.NB, l.WPS);
        (depth == 16 ? c->dk_block16 : c->dk_block32) = q->find(frag);
        const hc::BlockLaunch lm = hc::block_launch_config(a, c->mt_mini, &b);  // the short passes' variant (fewer tiles per workgroup)
        std::snprintf(frag, sizeof frag, "conv_block_kernelILi%dELi%dELi%dELi%dEEEv", lm.MT, lm.R, lm.NB, lm.WPS);
        (depth == 16 ? c->dk_mini16 : c->dk_mini32) = q->find(frag);
        if (depth == 16) {
            const hc::BlockLaunch ln = hc::block_launch_config(a, c->mt_narrow, &b);  // the narrow short pass: the 16-step kernel with its own tile count
            std::snprintf(frag, sizeof frag, "conv_block_kernelILi%dELi%dELi%dELi%dEEEv", ln.MT, ln.R, ln.NB, ln.WPS);
            c->dk_narrow = q->find(frag);
        }
    }
#ifdef HC_TUNING
    {   // the depth-64 pass of the tuning build (optional)
        hc::BlockArgs a{}, b{};
        a.depth   = 64;
        a.ngroups = 1;
        const hc::BlockLaunch l = hc::block_launch_config(a, c->mt_block64, &b);
        char frag[96];
        std::snprintf(frag, sizeof frag, "conv_block_kernelILi%dELi%dELi%dELi%dEEEv", l.MT, l.R, l.NB, l.WPS);
        c->dk_block64 = q->find(frag);
        if (c->dk_block64.kernarg != sizeof(hc::BlockArgs) || c->dk_block64.priv != 0) c->dk_block64 = hc::DirectKernel{};
    }
#endif
    if (!c->dk_narrow.ok() || c->dk_narrow.priv || c->dk_narrow.kernarg != sizeof(hc::BlockArgs)) {
        c->direct_why = "the narrow short-pass variant of the pass kernel is missing from hc_kernels.co";
        return;
    }
    if (!c->dk_mini16.ok() || !c->dk_mini32.ok() || c->dk_mini16.priv || c->dk_mini32.priv || c->dk_mini16.kernarg != sizeof(hc::BlockArgs) ||
        c->dk_mini32.kernarg != sizeof(hc::BlockArgs)) {
        c->direct_why = "the short-pass variant of the pass kernel is missing from hc_kernels.co";
        return;
    }
    if (!c->dk_finalize.ok() || !c->dk_scatter.ok() || !c->dk_reduce.ok() || !c->dk_block16.ok() || !c->dk_block32.ok() || !c->dk_near.ok()) {
        c->direct_why = "a kernel of this configuration is missing from hc_kernels.co";
        return;
    }
    // the code object must be the one built with this library: its kernels take exactly these argument blocks
    if (c->dk_finalize.kernarg != sizeof(hc::FinalizeArgs) || c->dk_scatter.kernarg != sizeof(hc::ScatterArgs) ||
        c->dk_block16.kernarg != sizeof(hc::BlockArgs) || c->dk_block32.kernarg != sizeof(hc::BlockArgs) || c->dk_reduce.kernarg != sizeof(hc::ReduceArgs) ||
        c->dk_near.kernarg != sizeof(hc::NearArgs)) {
        c->direct_why = "hc_kernels.co was not built from the same sources as this library (argument block sizes differ)";
        return;
    }
    if (c->dk_finalize.priv || c->dk_scatter.priv || c->dk_reduce.priv || c->dk_block16.priv || c->dk_block32.priv || c->dk_near.priv) {
        c->direct_why = "a kernel needs scratch memory";
        return;
    }
    if (!c->dk_added_mass.ok() || c->dk_added_mass.kernarg != sizeof(hc::AddedMassArgs) || c->dk_added_mass.priv != 0) {
        c->direct_why = "added_mass_mv_tagged_kernel is missing from hc_kernels.co";
        return;
    }
    static_assert(hc::kSlotArgBytes == hc::DirectQueue::kSlotBytes && hc::kSlotStateDoubles * sizeof(double) == hc::DirectQueue::kExtraBytes &&
                      12 * hc::kSlotStateMaxBodies + 1 <= hc::kSlotStateDoubles && 6 * hc::kSlotStateMaxBodies <= 1024,
                  "the state behind the step kernel's arguments: hc_limits.hpp and hc_direct.hpp must agree");
    static_assert(sizeof(hc::NearArgs) <= hc::DirectQueue::kSlotBytes && sizeof(hc::WideStepArgs) <= hc::DirectQueue::kSlotBytes,
                  "an argument block does not fit a kernarg slot of the direct queue");
    static_assert(sizeof(hc::ScatterArgs) <= hc::DirectQueue::kSlotBytes && sizeof(hc::FinalizeArgs) <= hc::DirectQueue::kSlotBytes &&
                      sizeof(hc::BlockArgs) <= hc::DirectQueue::kSlotBytes && sizeof(hc::StepArgs) <= hc::DirectQueue::kSlotBytes,
                  "an argument block does not fit a kernarg slot of the direct queue");
    // self-test 1: one dispatch of the reduction kernel with nothing to add must clear a marked word of P
    const double mark = 1.0;
    HC_HIP(hipMemcpy(c->d_P.p, &mark, sizeof mark, hipMemcpyHostToDevice));
    hc::ReduceArgs r{c->d_partials_block.p, 0, 0, c->Dpad, 1, c->d_P.p, c->d_E.p, c->d_err.p + 1, 0, 0, 0, 0};
    q->dispatch(c->dk_reduce, static_cast<uint32_t>((c->Dpad + 15) / 16), 256, 0, &r, sizeof r);
    if (!q->drain(2.0)) {
        c->direct_why = "self-test of the direct dispatch timed out";
        (void)q.release();  // a queue with a dispatch that never completed is left alone
        return;
    }
    double back = -1.0;
    HC_HIP(hipMemcpy(&back, c->d_P.p, sizeof back, hipMemcpyDeviceToHost));
    if (back != 0.0) { c->direct_why = "self-test of the direct dispatch failed"; return; }
    // self-test 2 (direct_selftest_rewrites): what the host RE-writes through the BAR must be what the next dispatch reads; lane 0
    // here, lane 1 when hc_added_mass_mv first uses it
    {
        bool abandon = false;
        if (!direct_selftest_rewrites(c, q.get(), 0, &abandon)) {
            if (abandon) (void)q.release();  // a queue with a dispatch that never completed is left alone
            return;
        }
    }
    c->dq           = q.release();
    c->direct_ready = true;
    c->direct_why.clear();
    {
        const int where = c->dq->ring_in_device_memory(0);
        c->direct_how   = where == 1 ? (c->dq->hdp_flush_available() ? "direct AQL dispatch (packet ring in device memory, HDP write-back in front of every doorbell)"
                                                                         : "direct AQL dispatch (packet ring in device memory, read-back in front of every doorbell: no HDP flush register)")
                          : where == 0 ? "direct AQL dispatch (packet ring in host memory; HC_QUEUE_DEV_MEM=1 moves the process's rings into device memory: "
                                         "1.0-1.8 us less per synchronous step, at a risk for the HIP runtime's own queues -- INTEGRATION.md)"
                                       : "direct AQL dispatch";
    }
    if (HC_TUNE_INT("HC_DEBUG_PLAN", 0) != 0) std::fprintf(stderr, "[hc] direct AQL dispatch in use for the step path\n");
}

}  // namespace detail
}  // namespace hc
