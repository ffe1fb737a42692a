// hc_direct.cpp -- see hc_direct.hpp.
#include "hc_direct.hpp"

#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <hsa/amd_hsa_signal.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <unistd.h>
#include <xmmintrin.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <stdexcept>
#include <vector>

namespace hc {

namespace {

struct KernelEntry {
    std::string name;
    DirectKernel k;
};

struct AgentPick {
    uint32_t want_bdf = 0, want_domain = 0;
    bool match_pci    = false;
    hsa_agent_t agent{};
    int found = 0;
};

hsa_status_t pick_agent(hsa_agent_t a, void* data) {
    AgentPick* p = static_cast<AgentPick*>(data);
    hsa_device_type_t type;
    if (hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &type) != HSA_STATUS_SUCCESS || type != HSA_DEVICE_TYPE_GPU) return HSA_STATUS_SUCCESS;
    if (p->match_pci) {
        uint32_t bdf = 0, domain = 0;
        if (hsa_agent_get_info(a, static_cast<hsa_agent_info_t>(HSA_AMD_AGENT_INFO_BDFID), &bdf) != HSA_STATUS_SUCCESS) return HSA_STATUS_SUCCESS;
        (void)hsa_agent_get_info(a, static_cast<hsa_agent_info_t>(HSA_AMD_AGENT_INFO_DOMAIN), &domain);
        if ((bdf & 0xFFF8u) != (p->want_bdf & 0xFFF8u) || domain != p->want_domain) return HSA_STATUS_SUCCESS;  // bus + device
    }
    if (p->found == 0) p->agent = a;
    p->found++;
    return HSA_STATUS_SUCCESS;
}

struct SymbolWalk {
    std::vector<KernelEntry>* out;
};

hsa_status_t walk_symbol(hsa_executable_t, hsa_agent_t, hsa_executable_symbol_t sym, void* data) {
    hsa_symbol_kind_t kind;
    if (hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_TYPE, &kind) != HSA_STATUS_SUCCESS || kind != HSA_SYMBOL_KIND_KERNEL)
        return HSA_STATUS_SUCCESS;
    uint32_t len = 0;
    if (hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_NAME_LENGTH, &len) != HSA_STATUS_SUCCESS || len == 0) return HSA_STATUS_SUCCESS;
    std::string name(len, '\0');
    if (hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_NAME, name.data()) != HSA_STATUS_SUCCESS) return HSA_STATUS_SUCCESS;
    KernelEntry e;
    e.name = name;
    (void)hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &e.k.object);
    (void)hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &e.k.group);
    (void)hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &e.k.priv);
    (void)hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE, &e.k.kernarg);
    static_cast<SymbolWalk*>(data)->out->push_back(e);
    return HSA_STATUS_SUCCESS;
}

const char* hsa_text(hsa_status_t s) {
    const char* m = nullptr;
    (void)hsa_status_string(s, &m);
    return m ? m : "unknown HSA error";
}

}  // namespace

struct DirectQueue::Impl {
    bool hsa_up = false;
    hsa_agent_t agent{};
    hsa_executable_t exe{};
    bool have_exe = false;
    hsa_code_object_reader_t reader{};
    bool have_reader = false;
    std::vector<char> image;
    std::vector<KernelEntry> kernels;
    static constexpr uint32_t kSlots = 64;
    struct Lane {
        std::atomic<int> error{0};  // hsa_status_t the runtime handed to the queue's error callback (0: none)
        hsa_queue_t* queue = nullptr;
        char* ring         = nullptr;  // kernarg slots: fine-grained device memory, written by the host through the BAR
        hsa_signal_t drain_sig{};
        bool have_drain_sig = false;
        hsa_signal_t gate{};  // what the parked barrier packet of arm() waits for (1: closed, 0: open)
        bool have_gate = false, armed = false;
        // Tuning experiment (HC_ARM_DEVICE_GATE, EXPERIMENTS.md round 6; null in the shipped library): the same gate once more, as a
        // signal record of our own in fine-grained DEVICE memory written through the BAR -- the runtime keeps its signals in host memory,
        // so a packet processor parked on `gate` polls across PCIe.
        volatile amd_signal_t* dev_gate = nullptr;
        bool ring_in_vram = false;  // the runtime put this lane's packet ring into device memory (ensure_lane)
    } lanes[DirectQueue::kLanes];
    char* ring_all = nullptr;
    // The GPU's HDP flush register (HSA_AMD_AGENT_INFO_HDP_FLUSH), or null.  What the host stores through the BAR -- kernel arguments,
    // body state and, with the packet ring in device memory, the AQL packets themselves -- passes the HDP on its way into VRAM, and a
    // doorbell takes another way into the chip: it can reach the packet processor while the tail of those stores is still in the HDP.
    // Harmless for arguments and state (the packet processor needs microseconds to start a wave; the per-step canary watches it), not
    // for the packet it reads at once: round 6's long differential runs died twice in ~2 000 cases with the ring in device memory and
    // no flush ("invalid code object", a memory fault at a garbage address: a packet whose header was new and whose second half was
    // old).  One store to this register in front of every doorbell writes the HDP back; stores to one device arrive in order.
    // HC_HDP_FLUSH=1 asks for that store in front of EVERY doorbell, also with the packet ring in host memory: arguments and state then
    // rest on the write-back instead of on the microseconds the packet processor takes (the HIP runtime guards its own device-side
    // kernel arguments the same way); 0.3-0.5 us per synchronous step (profiles/r06/host_path_c_hdp_flush.txt).
    volatile uint32_t* hdp_flush = nullptr;
    bool flush_always = false;  // HC_HDP_FLUSH=1
    uint64_t ticks_per_second = 0;
    struct Timed {
        hsa_signal_t sig;
        int tag;
        double aux;
    };
    std::vector<hsa_signal_t> free_sigs;
    std::vector<Timed> timed;
    std::vector<hsa_signal_t> order_sigs;  // ring of the cross-lane ordering signals (signal_after / wait_for)
    size_t order_next = 0;

    uint64_t reserve(Lane& ln) {
        hsa_queue_t* queue = ln.queue;
        const uint64_t idx = hsa_queue_add_write_index_relaxed(queue, 1);
        // room in the packet ring, and the kernarg slot of dispatch idx - kSlots is free: packets run in order, so once packet
        // j has been taken off the ring, packet j - 1 has completed
        // (a queue in the error state never advances: the waits end there and the packet written next is never run)
        uint64_t spins = 0;
        while (idx - hsa_queue_load_read_index_scacquire(queue) >= queue->size && !((++spins & 0xFFFF) == 0 && ln.error.load() != 0)) _mm_pause();
        while (idx >= kSlots && hsa_queue_load_read_index_scacquire(queue) + kSlots < idx + 2 && !((++spins & 0xFFFF) == 0 && ln.error.load() != 0)) _mm_pause();
        return idx;
    }
    static void open_gate(Lane& ln) {
        if (ln.dev_gate) {  // the record the packet processor polls locally first, then the runtime's
            ln.dev_gate->value = 0;
            _mm_sfence();
        }
        hsa_signal_store_screlease(ln.gate, 0);
    }
    void publish(Lane& ln, void* packet, uint16_t header, uint16_t setup, uint64_t idx) {
        hsa_queue_t* queue = ln.queue;
        // header + setup go last, in one 32-bit release store: the packet processor must not see a half-written packet
        const uint32_t word = static_cast<uint32_t>(header) | (static_cast<uint32_t>(setup) << 16);
        _mm_sfence();  // (a ring in device memory is write-combined memory: the packet's body leaves the core before its header)
        __atomic_store_n(reinterpret_cast<uint32_t*>(packet), word, __ATOMIC_RELEASE);
        _mm_sfence();
        if (flush_always && hdp_flush) {
            *hdp_flush = 1u;
        } else if (ln.ring_in_vram) {  // (a ring in host memory is read across PCIe, a microsecond behind the doorbell: what went through the BAR has landed)
            if (hdp_flush) *hdp_flush = 1u;  // everything stored through the BAR so far is in VRAM before the doorbell is acted on
            else (void)*reinterpret_cast<volatile uint32_t*>(packet);  // (no flush register mapped: a read through the BAR completes the
                                                                       // posted stores in front of it -- a PCIe round trip)
        }
        hsa_signal_store_screlease(queue->doorbell_signal, static_cast<hsa_signal_value_t>(idx));
    }
};

namespace {
// asynchronous queue errors arrive here on a runtime thread (the queue is then inactive: its packets never complete)
void on_queue_error(hsa_status_t status, hsa_queue_t*, void* data) {
    static_cast<std::atomic<int>*>(data)->store(status == HSA_STATUS_SUCCESS ? -1 : static_cast<int>(status), std::memory_order_release);
}
}  // namespace

namespace {
// 0 (what ships): the parked packet waits for the runtime's signal only.  Tuning build, HC_ARM_DEVICE_GATE=1: a barrier-OR over the
// runtime's signal and a signal record of our own in device memory; 2: a barrier-AND on the device-side record alone.  Measured in
// round 6 and not taken (EXPERIMENTS.md): the OR packet is slower than the plain one, not faster.
int use_device_gate() {
#ifdef HC_TUNING
    static const int mode = [] { const char* e = std::getenv("HC_ARM_DEVICE_GATE"); return e ? std::atoi(e) : 0; }();
    return mode;
#else
    return 0;
#endif
}
}  // namespace

DirectQueue::DirectQueue() : p_(new Impl) {}

bool DirectQueue::failed(int lane) const { return p_->lanes[lane].error.load(std::memory_order_acquire) != 0; }

std::string DirectQueue::failure_text() const {
    for (int l = 0; l < kLanes; ++l) {
        const int e = p_->lanes[l].error.load(std::memory_order_acquire);
        if (e != 0) return std::string(hsa_text(static_cast<hsa_status_t>(e))) + " (lane " + std::to_string(l) + ")";
    }
    return "no error";
}

DirectQueue::~DirectQueue() {
    Impl& p = *p_;
    for (int l = 0; l < kLanes; ++l) {
        if (p.lanes[l].queue) {
            if (p.lanes[l].armed) {  // open the gate: nothing may stay parked in a queue that is about to go
                Impl::open_gate(p.lanes[l]);
                p.lanes[l].armed = false;
                busy_[l]         = true;
            }
            if (busy_[l] && !failed(l)) (void)drain(5.0, l);
            (void)hsa_queue_destroy(p.lanes[l].queue);
        }
        if (p.lanes[l].have_drain_sig) (void)hsa_signal_destroy(p.lanes[l].drain_sig);
        if (p.lanes[l].have_gate) (void)hsa_signal_destroy(p.lanes[l].gate);
    }
    for (auto& t : p.timed) (void)hsa_signal_destroy(t.sig);
    for (auto& g : p.order_sigs) (void)hsa_signal_destroy(g);
    for (auto& s : p.free_sigs) (void)hsa_signal_destroy(s);
    if (p.ring_all) (void)hipFree(p.ring_all);
    if (p.have_exe) (void)hsa_executable_destroy(p.exe);
    if (p.have_reader) (void)hsa_code_object_reader_destroy(p.reader);
    if (p.hsa_up) (void)hsa_shut_down();
}

bool DirectQueue::init(int hip_device, const std::string& path, std::string* why) {
    Impl& p = *p_;
    auto fail = [&](const std::string& m) {
        if (why) *why = m;
        return false;
    };
    {
        std::ifstream f(path, std::ios::binary);
        if (!f) return fail("code object " + path + " not found");
        p.image.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
        if (p.image.size() < 64 || std::memcmp(p.image.data(), "\177ELF", 4) != 0) return fail(path + " is not an ELF code object");
    }
    hsa_status_t s = hsa_init();  // reference-counted: the HIP runtime holds the first reference
    if (s != HSA_STATUS_SUCCESS) return fail(std::string("hsa_init: ") + hsa_text(s));
    p.hsa_up = true;

    AgentPick pick;
    // always by PCI address: with HIP_VISIBLE_DEVICES set per rank HIP sees one device while HSA still enumerates every GPU
    int bus = 0, dev = 0, dom = 0;
    if (hipDeviceGetAttribute(&bus, hipDeviceAttributePciBusId, hip_device) == hipSuccess &&
        hipDeviceGetAttribute(&dev, hipDeviceAttributePciDeviceId, hip_device) == hipSuccess &&
        hipDeviceGetAttribute(&dom, hipDeviceAttributePciDomainID, hip_device) == hipSuccess) {
        pick.match_pci   = true;
        pick.want_bdf    = (static_cast<uint32_t>(bus) << 8) | (static_cast<uint32_t>(dev) << 3);
        pick.want_domain = static_cast<uint32_t>(dom);
    }
    (void)hipGetLastError();
    s = hsa_iterate_agents(pick_agent, &pick);
    if (s != HSA_STATUS_SUCCESS) return fail(std::string("hsa_iterate_agents: ") + hsa_text(s));
    if (pick.found == 0 && pick.match_pci) {
        // no agent reports this PCI address (a virtualised topology): unambiguous all the same if there is one GPU agent only
        pick = AgentPick{};
        s    = hsa_iterate_agents(pick_agent, &pick);
        if (s != HSA_STATUS_SUCCESS) return fail(std::string("hsa_iterate_agents: ") + hsa_text(s));
    }
    if (pick.found != 1) return fail("the HSA agent of the HIP device could not be identified");
    p.agent = pick.agent;

    s = hsa_code_object_reader_create_from_memory(p.image.data(), p.image.size(), &p.reader);
    if (s != HSA_STATUS_SUCCESS) return fail(std::string("code object reader: ") + hsa_text(s));
    p.have_reader = true;
    hsa_profile_t profile;
    s = hsa_agent_get_info(p.agent, HSA_AGENT_INFO_PROFILE, &profile);
    if (s != HSA_STATUS_SUCCESS) return fail(std::string("agent profile: ") + hsa_text(s));
    s = hsa_executable_create_alt(profile, HSA_DEFAULT_FLOAT_ROUNDING_MODE_DEFAULT, nullptr, &p.exe);
    if (s != HSA_STATUS_SUCCESS) return fail(std::string("executable: ") + hsa_text(s));
    p.have_exe = true;
    s = hsa_executable_load_agent_code_object(p.exe, p.agent, p.reader, nullptr, nullptr);
    if (s != HSA_STATUS_SUCCESS) return fail(std::string("loading the code object: ") + hsa_text(s));
    s = hsa_executable_freeze(p.exe, nullptr);
    if (s != HSA_STATUS_SUCCESS) return fail(std::string("freezing the executable: ") + hsa_text(s));
    SymbolWalk walk{&p.kernels};
    s = hsa_executable_iterate_agent_symbols(p.exe, p.agent, walk_symbol, &walk);
    if (s != HSA_STATUS_SUCCESS || p.kernels.empty()) return fail("no kernels in the code object");

    {
        hsa_amd_hdp_flush_t hdp{};
        if (hsa_agent_get_info(p.agent, static_cast<hsa_agent_info_t>(HSA_AMD_AGENT_INFO_HDP_FLUSH), &hdp) == HSA_STATUS_SUCCESS && hdp.HDP_MEM_FLUSH_CNTL)
            p.hdp_flush = hdp.HDP_MEM_FLUSH_CNTL;
        if (const char* e = std::getenv("HC_HDP_FLUSH")) p.flush_always = std::atoi(e) != 0;
    }
    // lane 0 (the step path) now; lane 1 (added-mass products) when it is first used (ensure_lane): a process that holds many
    // contexts on one device -- row shards sharing a GPU -- then keeps half as many hardware queues busy
    if (!ensure_lane(0, why)) return false;
    (void)hsa_amd_profiling_set_profiler_enabled(p.lanes[0].queue, 1);  // timestamps for the dispatches that carry a completion signal
    (void)hsa_system_get_info(HSA_SYSTEM_INFO_TIMESTAMP_FREQUENCY, &p.ticks_per_second);

    // kernarg ring in device memory the host can store into (same fault-free probe as BarBuffer, hc_runtime.cpp)
    void* q            = nullptr;
    const size_t ring_bytes = static_cast<size_t>(Impl::kSlots) * kSlotStride * kLanes;
    const size_t bytes      = ring_bytes + 4096;  // + one page for the lanes' device-side gate records
    if (hipExtMallocWithFlags(&q, bytes, hipDeviceMallocFinegrained) != hipSuccess) {
        (void)hipGetLastError();
        return fail("no fine-grained device memory for the kernel arguments");
    }
    p.ring_all = static_cast<char*>(q);
    for (int l = 0; l < kLanes; ++l) p.lanes[l].ring = p.ring_all + static_cast<size_t>(l) * Impl::kSlots * kSlotStride;
    bool host_ok = false;
    const int fz = open("/dev/zero", O_RDONLY), fn = open("/dev/null", O_WRONLY);
    if (fz >= 0 && fn >= 0)
        host_ok = read(fz, p.ring_all, bytes) == static_cast<ssize_t>(bytes) && write(fn, p.ring_all, bytes) == static_cast<ssize_t>(bytes);
    if (fz >= 0) close(fz);
    if (fn >= 0) close(fn);
    if (!host_ok) return fail("device memory is not host-addressable (no large BAR): kernel arguments cannot be stored directly");
    if (use_device_gate()) {
        for (int l = 0; l < kLanes; ++l) {
            volatile amd_signal_t* g = reinterpret_cast<volatile amd_signal_t*>(p.ring_all + ring_bytes + static_cast<size_t>(l) * 256);
            std::memset(const_cast<amd_signal_t*>(g), 0, sizeof(amd_signal_t));
            g->kind  = AMD_SIGNAL_KIND_USER;
            g->value = 0;
            p.lanes[l].dev_gate = g;
        }
        _mm_sfence();
    }
    return true;
}

bool DirectQueue::ensure_lane(int lane, std::string* why) {
    Impl& p = *p_;
    if (lane < 0 || lane >= kLanes) return false;
    if (p.lanes[lane].queue) return true;
    hsa_status_t s = hsa_queue_create(p.agent, 1024, HSA_QUEUE_TYPE_SINGLE, on_queue_error, &p.lanes[lane].error, UINT32_MAX, UINT32_MAX, &p.lanes[lane].queue);
    if (s != HSA_STATUS_SUCCESS) {
        p.lanes[lane].queue = nullptr;
        if (why) *why = std::string("hsa_queue_create: ") + hsa_text(s);
        return false;
    }
    if (!p.lanes[lane].have_drain_sig) {
        s = hsa_signal_create(1, 0, nullptr, &p.lanes[lane].drain_sig);
        if (s != HSA_STATUS_SUCCESS) {
            if (why) *why = std::string("hsa_signal_create: ") + hsa_text(s);
            return false;
        }
        p.lanes[lane].have_drain_sig = true;
    }
    if (!p.lanes[lane].have_gate && hsa_signal_create(1, 0, nullptr, &p.lanes[lane].gate) == HSA_STATUS_SUCCESS) p.lanes[lane].have_gate = true;
    p.lanes[lane].ring_in_vram = ring_in_device_memory(lane) == 1;
    return true;
}

bool DirectQueue::armed(int lane) const { return p_->lanes[lane].armed; }

void DirectQueue::arm(int lane) {
    Impl& p        = *p_;
    Impl::Lane& ln = p.lanes[lane];
    if (!ln.queue || !ln.have_gate || ln.armed || failed(lane)) return;
    hsa_signal_store_relaxed(ln.gate, 1);
    if (ln.dev_gate) {
        ln.dev_gate->value = 1;
        _mm_sfence();  // (in device memory before the packet that waits for it)
    }
    const uint64_t idx = p.reserve(ln);
    auto* pkt = reinterpret_cast<hsa_barrier_and_packet_t*>(ln.queue->base_address) + (idx & (ln.queue->size - 1));
    std::memset(reinterpret_cast<char*>(pkt) + 4, 0, sizeof(*pkt) - 4);
    // barrier-AND of one signal, or (tuning experiment) barrier-OR of the two gates: released by whichever opens first
    const bool dev_only = ln.dev_gate && use_device_gate() == 2;
    if (dev_only) pkt->dep_signal[0].handle = reinterpret_cast<uint64_t>(ln.dev_gate);
    else pkt->dep_signal[0] = ln.gate;
    if (ln.dev_gate && !dev_only) pkt->dep_signal[1].handle = reinterpret_cast<uint64_t>(ln.dev_gate);
    const uint16_t type   = (ln.dev_gate && !dev_only) ? HSA_PACKET_TYPE_BARRIER_OR : HSA_PACKET_TYPE_BARRIER_AND;
    const uint16_t header = (type << HSA_PACKET_HEADER_TYPE) | (1 << HSA_PACKET_HEADER_BARRIER);
    p.publish(ln, pkt, header, 0, idx);
    ln.armed = true;
}

DirectKernel DirectQueue::find(const std::string& fragment) const {
    DirectKernel hit;
    int n = 0;
    for (const auto& e : p_->kernels)
        if (e.name.find(fragment) != std::string::npos) {
            hit = e.k;
            ++n;
        }
    return n == 1 ? hit : DirectKernel{};
}

void DirectQueue::dispatch(const DirectKernel& k, uint32_t workgroups, uint32_t wg_size, uint32_t dyn_lds, const void* args, size_t arg_bytes,
                           int timed_tag, double timed_aux, int lane, FillExtra fill_extra, void* fill_user, bool no_acquire) {
    Impl& p            = *p_;
    Impl::Lane& ln     = p.lanes[lane];
    if (arg_bytes > kSlotBytes || k.kernarg > kSlotBytes || arg_bytes > std::max<size_t>(k.kernarg, 1))
        throw std::length_error("DirectQueue::dispatch: the argument block does not fit the kernel's kernarg segment / a ring slot");
    if (failed(lane)) throw std::runtime_error("DirectQueue::dispatch: the queue is in the error state: " + failure_text());
    const uint64_t idx = p.reserve(ln);
    char* slot         = ln.ring + (idx & (Impl::kSlots - 1)) * kSlotStride;
    std::memcpy(slot, args, arg_bytes);
    if (fill_extra) fill_extra(slot + kSlotBytes, fill_user);
    if (k.kernarg > arg_bytes) std::memset(slot + arg_bytes, 0, std::min<size_t>(k.kernarg, kSlotBytes) - arg_bytes);
    _mm_sfence();  // write-combined stores through the BAR are globally visible before the doorbell
    auto* pkt = reinterpret_cast<hsa_kernel_dispatch_packet_t*>(ln.queue->base_address) + (idx & (ln.queue->size - 1));
    pkt->workgroup_size_x     = static_cast<uint16_t>(wg_size);
    pkt->workgroup_size_y     = 1;
    pkt->workgroup_size_z     = 1;
    pkt->reserved0            = 0;
    pkt->grid_size_x          = workgroups * wg_size;
    pkt->grid_size_y          = 1;
    pkt->grid_size_z          = 1;
    pkt->private_segment_size = k.priv;
    pkt->group_segment_size   = k.group + dyn_lds;
    pkt->kernel_object        = k.object;
    pkt->kernarg_address      = slot;
    pkt->reserved2            = 0;
    pkt->completion_signal.handle = 0;
    if (timed_tag >= 0) {
        hsa_signal_t sig{};
        if (!p.free_sigs.empty()) {
            sig = p.free_sigs.back();
            p.free_sigs.pop_back();
            hsa_signal_store_relaxed(sig, 1);
        } else if (hsa_signal_create(1, 0, nullptr, &sig) != HSA_STATUS_SUCCESS) {
            sig.handle = 0;
        }
        if (sig.handle) {
            pkt->completion_signal = sig;
            p.timed.push_back({sig, timed_tag, timed_aux});
        }
    }
    // agent-scope fences: the inputs the host writes (state, arguments) live in fine-grained memory, which the GPU does not
    // cache, and results for the host leave through the end-of-kernel release
    const uint16_t header = (HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | (1 << HSA_PACKET_HEADER_BARRIER) |
                            ((no_acquire ? HSA_FENCE_SCOPE_NONE : HSA_FENCE_SCOPE_AGENT) << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) |
                            (HSA_FENCE_SCOPE_AGENT << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE);
    p.publish(ln, pkt, header, 1 << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS, idx);
    if (ln.armed) {  // the packet processor is parked on the barrier of arm(): open the gate now that the packet is in the queue
        Impl::open_gate(ln);
        ln.armed = false;
    }
    busy_[lane] = true;
}

bool DirectQueue::drain(double timeout_seconds, int lane) {
    Impl& p        = *p_;
    Impl::Lane& ln = p.lanes[lane];
    if (ln.queue && ln.armed) {  // a parked barrier would hold the drain packet back for ever
        Impl::open_gate(ln);
        ln.armed    = false;
        busy_[lane] = true;
    }
    if (!ln.queue || !busy_[lane]) return true;
    hsa_signal_store_relaxed(ln.drain_sig, 1);
    const uint64_t idx = p.reserve(ln);
    auto* pkt = reinterpret_cast<hsa_barrier_and_packet_t*>(ln.queue->base_address) + (idx & (ln.queue->size - 1));
    std::memset(reinterpret_cast<char*>(pkt) + 4, 0, sizeof(*pkt) - 4);
    pkt->completion_signal = ln.drain_sig;
    const uint16_t header  = (HSA_PACKET_TYPE_BARRIER_AND << HSA_PACKET_HEADER_TYPE) | (1 << HSA_PACKET_HEADER_BARRIER) |
                            (HSA_FENCE_SCOPE_SYSTEM << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) |
                            (HSA_FENCE_SCOPE_SYSTEM << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE);
    p.publish(ln, pkt, header, 0, idx);
    // active wait in slices of about a millisecond, so that the clock and the queue's error flag are looked at in between
    const double limit   = timeout_seconds > 0.0 ? timeout_seconds : 60.0;
    const uint64_t slice = std::max<uint64_t>(1, (p.ticks_per_second ? p.ticks_per_second : 100000000ull) / 1000);
    const auto t0        = std::chrono::steady_clock::now();
    while (hsa_signal_wait_scacquire(ln.drain_sig, HSA_SIGNAL_CONDITION_LT, 1, slice, HSA_WAIT_STATE_ACTIVE) >= 1) {
        if (failed(lane)) return false;
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit) return false;
    }
    busy_[lane] = false;
    return true;
}

void DirectQueue::abandon_lane(int lane) {
    Impl::Lane& ln = p_->lanes[lane];
    ln.queue       = nullptr;  // leaked on purpose
    ln.armed       = false;
    busy_[lane]    = false;
}

uint64_t DirectQueue::signal_after(int lane) {
    Impl& p        = *p_;
    Impl::Lane& ln = p.lanes[lane];
    if (!ln.queue || failed(lane)) return 0;
    if (p.order_sigs.empty()) {
        p.order_sigs.resize(kSignalRing);
        // only barrier packets wait for these (the host just looks at the value before it re-uses one): no interrupt on completion
        for (auto& g : p.order_sigs)
            if (hsa_amd_signal_create(0, 0, nullptr, HSA_AMD_SIGNAL_AMD_GPU_ONLY, &g) != HSA_STATUS_SUCCESS &&
                hsa_signal_create(0, 0, nullptr, &g) != HSA_STATUS_SUCCESS)
                g.handle = 0;
    }
    hsa_signal_t sig = p.order_sigs[p.order_next];
    p.order_next     = (p.order_next + 1) % p.order_sigs.size();
    if (!sig.handle) return 0;
    // its previous use lies kSignalRing calls back: long completed, or the queue is stuck (bounded look, then give the handle up)
    const uint64_t slice = std::max<uint64_t>(1, (p.ticks_per_second ? p.ticks_per_second : 100000000ull) / 1000);
    for (int tries = 0; hsa_signal_wait_scacquire(sig, HSA_SIGNAL_CONDITION_LT, 1, slice, HSA_WAIT_STATE_ACTIVE) >= 1; ++tries)
        if (tries > 2000 || failed(lane)) return 0;
    hsa_signal_store_relaxed(sig, 1);
    const uint64_t idx = p.reserve(ln);
    auto* pkt = reinterpret_cast<hsa_barrier_and_packet_t*>(ln.queue->base_address) + (idx & (ln.queue->size - 1));
    std::memset(reinterpret_cast<char*>(pkt) + 4, 0, sizeof(*pkt) - 4);
    pkt->completion_signal = sig;
    const uint16_t header  = (HSA_PACKET_TYPE_BARRIER_AND << HSA_PACKET_HEADER_TYPE) | (1 << HSA_PACKET_HEADER_BARRIER) |
                            (HSA_FENCE_SCOPE_AGENT << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) |
                            (HSA_FENCE_SCOPE_AGENT << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE);
    p.publish(ln, pkt, header, 0, idx);
    if (ln.armed) {
        Impl::open_gate(ln);
        ln.armed = false;
    }
    busy_[lane] = true;
    return sig.handle;
}

void DirectQueue::wait_for(int lane, uint64_t handle) {
    Impl& p        = *p_;
    Impl::Lane& ln = p.lanes[lane];
    if (!ln.queue || failed(lane) || handle == 0) return;
    const uint64_t idx = p.reserve(ln);
    auto* pkt = reinterpret_cast<hsa_barrier_and_packet_t*>(ln.queue->base_address) + (idx & (ln.queue->size - 1));
    std::memset(reinterpret_cast<char*>(pkt) + 4, 0, sizeof(*pkt) - 4);
    pkt->dep_signal[0].handle = handle;
    const uint16_t header     = (HSA_PACKET_TYPE_BARRIER_AND << HSA_PACKET_HEADER_TYPE) | (1 << HSA_PACKET_HEADER_BARRIER) |
                            (HSA_FENCE_SCOPE_AGENT << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) |
                            (HSA_FENCE_SCOPE_AGENT << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE);
    p.publish(ln, pkt, header, 0, idx);
    if (ln.armed) {
        Impl::open_gate(ln);
        ln.armed = false;
    }
    busy_[lane] = true;
}

void DirectQueue::enable_timing(int lane) {
    if (p_->lanes[lane].queue) (void)hsa_amd_profiling_set_profiler_enabled(p_->lanes[lane].queue, 1);
}

uint32_t DirectQueue::compute_units() const {
    uint32_t ncu = 0;
    if (hsa_agent_get_info(p_->agent, static_cast<hsa_agent_info_t>(HSA_AMD_AGENT_INFO_COMPUTE_UNIT_COUNT), &ncu) != HSA_STATUS_SUCCESS) return 0;
    return ncu;
}

bool DirectQueue::set_cu_mask(int lane, uint32_t keep) {
    Impl::Lane& ln     = p_->lanes[lane];
    const uint32_t ncu = compute_units();
    if (!ln.queue || ncu == 0 || keep == 0 || keep > ncu) return false;
    std::vector<uint32_t> mask((ncu + 31) / 32, 0u);
    for (uint32_t i = 0; i < keep; ++i) mask[i / 32] |= 1u << (i % 32);
    return hsa_amd_queue_cu_set_mask(ln.queue, static_cast<uint32_t>(mask.size() * 32), mask.data()) == HSA_STATUS_SUCCESS;
}

size_t DirectQueue::timed_pending() const { return p_->timed.size(); }

uint64_t DirectQueue::system_ticks() const {
    uint64_t t = 0;
    (void)hsa_system_get_info(HSA_SYSTEM_INFO_TIMESTAMP, &t);
    return t;
}
uint64_t DirectQueue::gpu_to_system(uint64_t gpu_ticks) const {
    uint64_t t = 0;
    if (hsa_amd_profiling_convert_tick_to_system_domain(p_->agent, gpu_ticks, &t) != HSA_STATUS_SUCCESS) return 0;
    return t;
}
uint64_t DirectQueue::system_ticks_per_second() const { return p_->ticks_per_second; }

bool DirectQueue::hdp_flush_available() const { return p_->hdp_flush != nullptr; }

int DirectQueue::ring_in_device_memory(int lane) const {
    const Impl::Lane& ln = p_->lanes[lane];
    if (!ln.queue || !ln.queue->base_address) return -1;
    hsa_amd_pointer_info_t info{};
    info.size = sizeof(info);
    if (hsa_amd_pointer_info(ln.queue->base_address, &info, nullptr, nullptr, nullptr) != HSA_STATUS_SUCCESS) return -1;
    if (info.type != HSA_EXT_POINTER_TYPE_HSA) return -1;
    return info.agentOwner.handle == p_->agent.handle ? 1 : 0;
}

void DirectQueue::collect(const std::function<void(int, double, double)>& sink) {
    Impl& p = *p_;
    for (auto& t : p.timed) {
        const uint64_t slice = std::max<uint64_t>(1, (p.ticks_per_second ? p.ticks_per_second : 100000000ull) / 1000);
        const auto t0        = std::chrono::steady_clock::now();
        bool done            = true;
        while (hsa_signal_wait_scacquire(t.sig, HSA_SIGNAL_CONDITION_LT, 1, slice, HSA_WAIT_STATE_ACTIVE) >= 1) {
            if (failed(0) || std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 60.0) { done = false; break; }
        }
        if (!done) continue;  // the signal is not recycled: its dispatch never completed
        hsa_amd_profiling_dispatch_time_t dt{};
        if (hsa_amd_profiling_get_dispatch_time(p.agent, t.sig, &dt) == HSA_STATUS_SUCCESS && p.ticks_per_second > 0 && dt.end >= dt.start)
            sink(t.tag, static_cast<double>(dt.end - dt.start) / static_cast<double>(p.ticks_per_second), t.aux);
        p.free_sigs.push_back(t.sig);
    }
    p.timed.clear();
}

}  // namespace hc
