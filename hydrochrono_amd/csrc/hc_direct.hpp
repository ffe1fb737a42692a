// hc_direct.hpp -- kernel dispatch without the HIP launch call: AQL packets written straight into an HSA queue of our own.
//
// hipLaunchKernelGGL costs the host 2.4-3.3 us per call on this stack, and the synchronous step (hc_step) pays it on its
// critical path once per evaluation.  An AQL packet + kernel arguments + doorbell written by hand costs 0.15-0.25 us, and with
// the arguments in device memory (written through the PCIe BAR) and agent-scope fences the launch-to-result time of a small
// kernel drops from 7.9-9.8 us to 5.6-7.5 us (profiles/r02/latency_probe5.cpp).  The kernels are the same code: the stand-alone
// code object hc_kernels.co is built from hc_kernels.hip next to the library and loaded through the HSA loader.
//
// Packets of one queue with the barrier bit execute in order, like kernels of a HIP stream; nothing orders this queue against
// HIP streams, so the owner drains one side before it switches to the other (hc_step.cpp).
#pragma once
#include <cstddef>
#include <cstdint>
#include <functional>
#include <memory>
#include <string>

namespace hc {

struct DirectKernel {
    uint64_t object  = 0;  // kernel descriptor address
    uint32_t group   = 0;  // static LDS bytes
    uint32_t priv    = 0;  // scratch bytes per work-item
    uint32_t kernarg = 0;  // kernarg segment bytes
    bool ok() const { return object != 0; }
};

class DirectQueue {
  public:
    DirectQueue();
    ~DirectQueue();
    DirectQueue(const DirectQueue&)            = delete;
    DirectQueue& operator=(const DirectQueue&) = delete;

    // Binds to the HSA agent of HIP device `hip_device`, loads the code object, creates the queue and the kernarg ring.
    // false (with the reason in *why): the direct path is not available and the caller keeps using HIP launches.
    bool init(int hip_device, const std::string& code_object_path, std::string* why);
    // Creates the lane's HSA queue if it does not exist yet (init creates lane 0 only).  false: the lane cannot be used.
    bool ensure_lane(int lane, std::string* why);
    // Kernel whose mangled name contains `fragment` (must match exactly one kernel); !ok() if there is none.
    DirectKernel find(const std::string& fragment) const;

    // One kernel dispatch: grid of `workgroups` x `wg_size` work-items, `dyn_lds` bytes of dynamic LDS, the argument block
    // copied into the next kernarg slot.  timed >= 0: the dispatch carries a completion signal and its device-side duration
    // is reported by collect() with this tag.
    // lane: which of the object's independent queues (0: the step path; 1: the added-mass product, which must never wait behind
    // work the step path still runs; 2: look-ahead passes that run beside the steps, see signal_after / wait_for / set_cu_mask).
    // fill_extra: called with the host address of the kExtraBytes that FOLLOW the slot's argument block (same memory, same store path);
    // a kernel built for it finds them at its kernarg segment pointer + kSlotBytes -- an address it knows before it has loaded a single
    // argument, so it can request what the host put there together with its arguments instead of after them (the step kernel's body
    // state, hc_step.cpp).
    using FillExtra = void (*)(char* extra, void* user);
    // no_acquire (tuning experiment, EXPERIMENTS.md round 6): the packet carries no acquire fence (the caches are not invalidated in
    // front of the kernel); the release fence stays.
    void dispatch(const DirectKernel& k, uint32_t workgroups, uint32_t wg_size, uint32_t dyn_lds, const void* args, size_t arg_bytes,
                  int timed_tag = -1, double timed_aux = 0.0, int lane = 0, FillExtra fill_extra = nullptr, void* fill_user = nullptr,
                  bool no_acquire = false);
    // Parks the lane's packet processor on a barrier packet that waits for a signal; the next dispatch() releases it right after
    // its packet is in the queue.  A queue that has sat EMPTY for more than a few tens of microseconds takes about 6 us longer from
    // doorbell to kernel start (12.2 against 6.0 us launch-to-result for a small kernel after >= 100 us of idle,
    // profiles/r03/latency_probe6.cpp -- a Chrono loop leaves such gaps between force evaluations); a parked queue does not
    // (6.2 us after any gap).  Right before a dispatch that follows within microseconds the extra packet costs 1.7 us, so the owner
    // arms only when it expects the caller to be away for a while.
    void arm(int lane = 0);
    bool armed(int lane = 0) const;
    // Waits until everything dispatched to the lane so far has completed.  Gives up after timeout_seconds (<= 0: one minute), or
    // as soon as the queue has reported an error, and returns false (the queue must then not be used any more).
    bool drain(double timeout_seconds = 0.0, int lane = 0);
    // The HSA runtime has reported an asynchronous error on the lane's queue (bad packet, memory fault, ...): nothing dispatched
    // to it will complete any more.
    bool failed(int lane = 0) const;
    std::string failure_text() const;
    bool busy(int lane = 0) const { return busy_[lane]; }
    // A lane whose self-test dispatch never completed: its HSA queue is forgotten (not drained, not destroyed -- both would wait for
    // the dispatch) and the lane reads as idle from now on.
    void abandon_lane(int lane);
    static constexpr int kLanes = 3;  // 0: the step path; 1: added-mass products; 2: look-ahead passes that run beside the steps
    // Cross-lane ordering.  signal_after(lane): a barrier packet behind everything dispatched to the lane so far; the returned
    // handle completes when all of that has finished (0: no signal could be had -- the caller must not rely on ordering then).
    // wait_for(lane, handle): a barrier packet that holds back everything dispatched to the lane AFTER it until the handle has
    // completed.  Handles come from a small ring of signals and stay valid for the next kSignalRing - 1 calls of signal_after.
    uint64_t signal_after(int lane);
    void wait_for(int lane, uint64_t handle);
    static constexpr int kSignalRing = 64;
    // Restricts the lane's queue to the first `keep` compute units in the runtime's mask order (bits are dealt to the XCDs in turn,
    // so every XCD keeps the same number free for the other lanes).  false: the runtime refused.
    bool set_cu_mask(int lane, uint32_t keep);
    void enable_timing(int lane);  // device timestamps for the lane's timed dispatches (lane 0 has them from init)
    uint32_t compute_units() const;
    // Reports (tag, seconds, aux) of every timed dispatch since the last call (waits for them).
    void collect(const std::function<void(int, double, double)>& sink);
    size_t timed_pending() const;
    // Clocks (diagnostics: the step kernel's stage clock, hc_tuning_step_stamps).  system_ticks(): the HSA system timestamp now;
    // gpu_to_system(): a value of the GPU's constant clock (s_memrealtime) in that domain; both in units of 1 / system_ticks_per_second().
    uint64_t system_ticks() const;
    uint64_t gpu_to_system(uint64_t gpu_ticks) const;
    uint64_t system_ticks_per_second() const;
    // Where the runtime put the lane's AQL packet ring: 1 device memory (the packet processor fetches packets locally: 1.4-1.9 us less
    // from doorbell to kernel start than over PCIe, profiles/r06/queue_dev_mem_ab.txt), 0 host memory, -1 unknown.  The runtime decides
    // when it is initialised (HSA_ALLOCATE_QUEUE_DEV_MEM, requested by this library at load time -- hc_runtime.cpp).
    int ring_in_device_memory(int lane = 0) const;
    bool hdp_flush_available() const;  // the HDP flush register is mapped: every doorbell is preceded by a write-back of the HDP

    static constexpr size_t kSlotBytes  = 4096;   // kernarg bytes per dispatch (the largest argument block is the scatter's 2.6 KB)
    static constexpr size_t kExtraBytes = 16384;  // ... followed by room for data the kernel addresses relative to its kernarg pointer
    static constexpr size_t kSlotStride = kSlotBytes + kExtraBytes;

  private:
    struct Impl;
    std::unique_ptr<Impl> p_;
    bool busy_[kLanes] = {false, false, false};
};

}  // namespace hc
