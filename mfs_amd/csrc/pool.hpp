// pool.hpp -- library-owned staging memory for the host-pointer entry points ("Ownership": the
// caller owns every buffer it passes; the library stages through its own device buffers and pinned pool, reused across
// calls).  Host-only code.
//
//   BlockPool<Device>  cached hipMalloc blocks, per device
//   BlockPool<Pinned>  cached hipHostMalloc blocks (page-locked host memory), per device context
//   CallContext        a compute stream, a copy stream and a few events, recycled between calls
//
// Nothing here is freed on the hot path: a block goes back to its pool when the call that borrowed it returns, and a
// later call of the same (or a smaller) size takes it again.  mfs_pool_trim() gives everything unused back to the driver.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <mutex>
#include <vector>

namespace mfs {

constexpr int kMaxDevices = 64;

struct PoolCounters {
    uint64_t cached_bytes = 0;   // bytes currently held by the pool (in use or free)
    uint64_t fresh_allocs = 0;   // driver allocations made so far (a steady-state call makes none)
    uint64_t acquires = 0;
};

template <bool Pinned>
class BlockPool {
public:
    // smallest cached free block with size in [bytes, 2 * bytes + slack]; a fresh driver allocation otherwise
    hipError_t acquire(void** out, size_t bytes) {
        *out = nullptr;
        if (bytes == 0) bytes = 8;
        bytes = (bytes + 255) & ~size_t(255);
        {
            std::lock_guard<std::mutex> lk(mu_);
            ++counters_.acquires;
            int best = -1;
            for (int i = 0; i < (int)blocks_.size(); ++i) {
                const Block& b = blocks_[i];
                if (!b.in_use && b.bytes >= bytes && b.bytes <= 2 * bytes + (1u << 20) &&
                    (best < 0 || b.bytes < blocks_[best].bytes)) best = i;
            }
            if (best >= 0) { blocks_[best].in_use = true; *out = blocks_[best].ptr; return hipSuccess; }
        }
        void* p = nullptr;
        hipError_t e = Pinned ? hipHostMalloc(&p, bytes, hipHostMallocDefault) : hipMalloc(&p, bytes);
        if (e != hipSuccess) {           // out of memory with blocks parked in the cache: give them back and retry once
            (void)hipGetLastError();
            trim();
            e = Pinned ? hipHostMalloc(&p, bytes, hipHostMallocDefault) : hipMalloc(&p, bytes);
            if (e != hipSuccess) return e;
        }
        std::lock_guard<std::mutex> lk(mu_);
        blocks_.push_back(Block{p, bytes, true});
        counters_.cached_bytes += bytes;
        ++counters_.fresh_allocs;
        *out = p;
        return hipSuccess;
    }
    // returns false when `p` is not one of this pool's blocks
    bool release(void* p) {
        if (!p) return true;
        std::lock_guard<std::mutex> lk(mu_);
        for (Block& b : blocks_)
            if (b.ptr == p) { b.in_use = false; return true; }
        return false;
    }
    void trim() {
        std::vector<Block> drop;
        {
            std::lock_guard<std::mutex> lk(mu_);
            std::vector<Block> keep;
            for (const Block& b : blocks_) (b.in_use ? keep : drop).push_back(b);
            blocks_.swap(keep);
            for (const Block& b : drop) counters_.cached_bytes -= b.bytes;
        }
        for (const Block& b : drop) { if (Pinned) (void)hipHostFree(b.ptr); else (void)hipFree(b.ptr); }
    }
    PoolCounters counters() {
        std::lock_guard<std::mutex> lk(mu_);
        return counters_;
    }

private:
    struct Block { void* ptr; size_t bytes; bool in_use; };
    std::mutex mu_;
    std::vector<Block> blocks_;
    PoolCounters counters_;
};

struct CallContext {
    hipStream_t compute = nullptr, copy = nullptr;
    static constexpr int kEvents = 34;
    hipEvent_t ev[kEvents] = {nullptr};
};

class ContextPool {
public:
    hipError_t acquire(CallContext** out) {
        {
            std::lock_guard<std::mutex> lk(mu_);
            if (!free_.empty()) { *out = free_.back(); free_.pop_back(); return hipSuccess; }
        }
        CallContext* c = new CallContext();
        hipError_t e = hipStreamCreateWithFlags(&c->compute, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->copy, hipStreamNonBlocking);
        for (int i = 0; i < CallContext::kEvents && e == hipSuccess; ++i)
            e = hipEventCreateWithFlags(&c->ev[i], hipEventDisableTiming);
        if (e != hipSuccess) { destroy(c); return e; }
        *out = c;
        return hipSuccess;
    }
    void release(CallContext* c) {
        if (!c) return;
        std::lock_guard<std::mutex> lk(mu_);
        free_.push_back(c);
    }
    void trim() {
        std::vector<CallContext*> drop;
        { std::lock_guard<std::mutex> lk(mu_); drop.swap(free_); }
        for (CallContext* c : drop) destroy(c);
    }

private:
    static void destroy(CallContext* c) {
        for (hipEvent_t e : c->ev) if (e) (void)hipEventDestroy(e);
        if (c->copy) (void)hipStreamDestroy(c->copy);
        if (c->compute) (void)hipStreamDestroy(c->compute);
        delete c;
    }
    std::mutex mu_;
    std::vector<CallContext*> free_;
};

struct DeviceState {
    BlockPool<false> device;
    BlockPool<true> pinned;
    ContextPool contexts;
};

inline DeviceState& device_state(int device) {
    static DeviceState states[kMaxDevices];
    return states[(device >= 0 && device < kMaxDevices) ? device : 0];
}

// A set of pool blocks borrowed for the duration of one call.
class Lease {
public:
    explicit Lease(int device) : st_(device_state(device)) {}
    ~Lease() {
        for (void* p : dev_) st_.device.release(p);
        for (void* p : pin_) st_.pinned.release(p);
        st_.contexts.release(ctx_);
    }
    Lease(const Lease&) = delete;
    Lease& operator=(const Lease&) = delete;
    template <typename T>
    hipError_t device_block(T** out, size_t bytes) {
        void* p = nullptr;
        hipError_t e = st_.device.acquire(&p, bytes);
        if (e == hipSuccess) dev_.push_back(p);
        *out = static_cast<T*>(p);
        return e;
    }
    template <typename T>
    hipError_t pinned_block(T** out, size_t bytes) {
        void* p = nullptr;
        hipError_t e = st_.pinned.acquire(&p, bytes);
        if (e == hipSuccess) pin_.push_back(p);
        *out = static_cast<T*>(p);
        return e;
    }
    hipError_t context(CallContext** out) {
        if (!ctx_) { hipError_t e = st_.contexts.acquire(&ctx_); if (e != hipSuccess) return e; }
        *out = ctx_;
        return hipSuccess;
    }

private:
    DeviceState& st_;
    std::vector<void*> dev_, pin_;
    CallContext* ctx_ = nullptr;
};

}  // namespace mfs
