// Device memory, streams, events and pinned words of libsnekmer_hip.so: everything the HIP runtime hands out is taken
// ONCE and recycled here.
//
// Why ("The stop"): until round 5 every skm_malloc was a hipMalloc and every skm_free a hipFree, a
// scratch slot that grew was hipFree'd, and every side context of engine.OverlappedPipeline created and destroyed a
// CU-masked stream (a hardware queue of its own), a pinned page and a handful of events.  hipFree waits, implicitly, for
// every stream of the device; callers (score._set_measure) let arrays go while the kernels reading them were still
// queued and relied on exactly that wait.  Three long fuzz runs stopped inside such a hipFree.  Nothing in this library
// rests on that wait any more:
//   * skm_free parks the block with one event per stream of the device that is not idle at that moment (skm_ctx registry);
//     skm_malloc reuses a parked block of the same size class only once all of its events have completed, and otherwise
//     takes new memory.  Reuse is therefore ordered behind every kernel that could have touched the block, on any
//     stream, without a host wait and without the allocating stream having to wait either.
//   * hipFree is only called when the device has run out of memory (or skm_mem_trim asks), and then AFTER an explicit
//     hipStreamSynchronize of every registered stream: the runtime's implicit wait finds an idle device.
//   * streams (plain and CU-masked), pinned pages and events go back to per-device caches when a context is destroyed:
//     after warm-up, creating and destroying contexts creates and destroys nothing in the runtime.
// SKM_GUARD=1 (environment, read once): 512 bytes of 0xA5 behind every array, checked when it is freed: a kernel that
// writes past the end of an array is reported by skm_free (SKM_E_HIP) with the array's size and the first bad offset.
// skm_debug_report: what a watchdog prints when a call does not return (which stream is busy, which timed kernel has
// not finished, what the pool holds).
#include <algorithm>
#include <chrono>
#include <map>
#include <mutex>
#include <unordered_map>

#include "skm_common.h"

namespace {

constexpr size_t GUARD_BYTES = 512;
constexpr unsigned char GUARD_BYTE = 0xA5;
constexpr int MAX_DEVICES = 64;

struct parked_block {
    void *ptr = nullptr;
    std::vector<hipEvent_t> pending;  // one per stream that was busy when the block was freed
};

struct live_block {
    size_t class_bytes = 0, user_bytes = 0;
    bool guarded = false;
};

struct cached_stream {
    int first = -1, last = -1;  // CU groups, -1: unmasked
    hipStream_t stream = nullptr;
};

struct device_pool {
    std::timed_mutex mu;
    std::multimap<size_t, parked_block> parked;   // by size class
    std::vector<std::pair<void *, size_t>> limbo;  // freed while a capture was open: parked once no capture is
    std::unordered_map<void *, live_block> live;
    std::vector<skm_ctx *> contexts;
    std::vector<cached_stream> streams;
    std::vector<void *> pinned_pages;
    std::vector<hipEvent_t> events[2];  // [0] hipEventDisableTiming, [1] timing
    int64_t live_bytes = 0, parked_bytes = 0, n_malloc = 0, n_free = 0, n_reuse = 0, n_trim = 0, n_wait_skipped = 0;
};

device_pool *g_pools[MAX_DEVICES] = {};
std::mutex g_pools_mu;
int g_guard = -1;
bool g_exit_hook = false, g_shut_down = false;

void shutdown_at_exit();

// nullptr once the process is exiting (the callers then hand their objects straight back to the runtime)
device_pool *pool_of(int device)
{
    if (device < 0 || device >= MAX_DEVICES)
        return nullptr;
    std::lock_guard<std::mutex> lock(g_pools_mu);
    if (g_shut_down)
        return nullptr;
    if (!g_exit_hook) {
        // registered at the first use of the library, i.e. after the HIP runtime registered its own exit handlers: runs
        // BEFORE them, while streams and memory can still be handed back in an orderly way (a profiler that walks the
        // process's queues in its finaliser crashed on the cached CU-masked streams: rocprofv3 --pmc, ROCm 7.2)
        g_exit_hook = true;
        atexit(shutdown_at_exit);
    }
    if (!g_pools[device])
        g_pools[device] = new device_pool();  // never deleted: its mutex may be in use by a thread that outlives main()
    return g_pools[device];
}

bool guard_on()
{
    if (g_guard < 0) {
        const char *e = getenv("SKM_GUARD");
        g_guard = (e && *e && strcmp(e, "0") != 0) ? 1 : 0;
    }
    return g_guard == 1;
}

// Size classes: 4 KiB granules up to 64 KiB, then eight steps per octave (at most 12.5 % over) up to 1 GiB, 2 MiB granules
// above (the large arrays of a job recur with exactly the same size: the output block, the dense count matrix).
size_t size_class(size_t bytes)
{
    if (bytes <= (size_t)64 << 10)
        return (bytes + 4095) & ~(size_t)4095;
    if (bytes >= (size_t)1 << 30)
        return (bytes + (((size_t)2 << 20) - 1)) & ~(((size_t)2 << 20) - 1);
    int top = 63 - __builtin_clzll((unsigned long long)bytes);
    const size_t step = (size_t)1 << (top - 3);
    return (bytes + step - 1) & ~(step - 1);
}

hipEvent_t take_event(device_pool *p, int timing)
{
    if (!p->events[timing].empty()) {
        hipEvent_t e = p->events[timing].back();
        p->events[timing].pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    const hipError_t rc = timing ? hipEventCreate(&e) : hipEventCreateWithFlags(&e, hipEventDisableTiming);
    if (rc != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return e;
}

// the streams a context may have work on
template <typename F>
void for_streams(const skm_ctx *c, F &&f)
{
    if (c->stream)
        f(c->stream);
    if (c->s_writer)
        f(c->s_writer);
    if (c->s_gram)
        f(c->s_gram);
}

bool any_capture(const device_pool *p)
{
    for (const skm_ctx *c : p->contexts)
        if (c->capturing)
            return true;
    return false;
}

// park `ptr`: an event on every stream that still has work queued (pool lock held, no capture open)
void park_locked(device_pool *p, void *ptr, size_t cls)
{
    parked_block b;
    b.ptr = ptr;
    bool lost = false;
    for (const skm_ctx *c : p->contexts)
        for_streams(c, [&](hipStream_t s) {
            if (hipStreamQuery(s) == hipSuccess)
                return;  // idle: nothing queued there can still touch the block
            (void)hipGetLastError();
            hipEvent_t e = take_event(p, 0);
            if (!e || hipEventRecord(e, s) != hipSuccess) {
                (void)hipGetLastError();
                if (e)
                    p->events[0].push_back(e);
                lost = true;  // cannot mark that stream: wait for it instead (rare: out of events)
                (void)hipStreamSynchronize(s);
                return;
            }
            b.pending.push_back(e);
        });
    (void)lost;
    p->parked.emplace(cls, std::move(b));
    p->parked_bytes += (int64_t)cls;
}

void drain_limbo_locked(device_pool *p)
{
    if (p->limbo.empty() || any_capture(p))
        return;
    for (auto &it : p->limbo)
        park_locked(p, it.first, it.second);
    p->limbo.clear();
}

bool block_idle(device_pool *p, parked_block &b)
{
    while (!b.pending.empty()) {
        if (hipEventQuery(b.pending.back()) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        p->events[0].push_back(b.pending.back());
        b.pending.pop_back();
    }
    return true;
}

// explicit wait for every registered stream of the device (never a capturing one: that would invalidate the capture)
void quiesce_locked(device_pool *p)
{
    for (const skm_ctx *c : p->contexts) {
        if (c->capturing)
            continue;
        for_streams(c, [&](hipStream_t s) {
            if (hipStreamSynchronize(s) != hipSuccess)
                (void)hipGetLastError();
        });
    }
}

// give every parked block back to the runtime (the device is made idle first, explicitly)
int64_t trim_locked(device_pool *p)
{
    if (p->parked.empty())
        return 0;
    quiesce_locked(p);
    int64_t released = 0;
    for (auto it = p->parked.begin(); it != p->parked.end();) {
        if (!block_idle(p, it->second)) {  // (only behind a capturing context's stream)
            ++it;
            continue;
        }
        if (hipFree(it->second.ptr) != hipSuccess)
            (void)hipGetLastError();
        ++p->n_free;
        released += (int64_t)it->first;
        p->parked_bytes -= (int64_t)it->first;
        it = p->parked.erase(it);
    }
    ++p->n_trim;
    return released;
}

void shutdown_at_exit()
{
    device_pool *pools[MAX_DEVICES];
    {
        std::lock_guard<std::mutex> lock(g_pools_mu);
        g_shut_down = true;
        memcpy(pools, g_pools, sizeof(pools));
    }
    for (int d = 0; d < MAX_DEVICES; ++d) {
        device_pool *p = pools[d];
        if (!p || !p->mu.try_lock_for(std::chrono::seconds(2)))
            continue;
        if (hipSetDevice(d) == hipSuccess) {
            quiesce_locked(p);  // contexts the program never destroyed
            for (auto &it : p->limbo)
                (void)hipFree(it.first);
            p->limbo.clear();
            for (auto &it : p->parked) {
                for (hipEvent_t e : it.second.pending)
                    (void)hipEventDestroy(e);
                (void)hipFree(it.second.ptr);
            }
            p->parked.clear();
            p->parked_bytes = 0;
            for (auto &cs : p->streams)
                (void)hipStreamDestroy(cs.stream);
            p->streams.clear();
            for (int t = 0; t < 2; ++t) {
                for (hipEvent_t e : p->events[t])
                    (void)hipEventDestroy(e);
                p->events[t].clear();
            }
            for (void *page : p->pinned_pages)
                (void)hipHostFree(page);
            p->pinned_pages.clear();
        }
        (void)hipGetLastError();
        p->mu.unlock();
    }
}

}  // namespace

// ---------------------------------------------------------------------------- registry
void skm_registry_add(skm_ctx *ctx)
{
    device_pool *p = pool_of(ctx->device);
    if (!p)
        return;
    std::lock_guard<std::timed_mutex> lock(p->mu);
    p->contexts.push_back(ctx);
}

void skm_registry_remove(skm_ctx *ctx)
{
    device_pool *p = pool_of(ctx->device);
    if (!p)
        return;
    std::lock_guard<std::timed_mutex> lock(p->mu);
    p->contexts.erase(std::remove(p->contexts.begin(), p->contexts.end(), ctx), p->contexts.end());
}

int skm_quiesce_device(int device)
{
    device_pool *p = pool_of(device);
    if (!p)
        return SKM_OK;
    std::lock_guard<std::timed_mutex> lock(p->mu);
    quiesce_locked(p);
    return SKM_OK;
}

// ---------------------------------------------------------------------------- streams, events, pinned words
hipError_t skm_stream_acquire(int device, int first_group, int last_group, const uint32_t *mask, uint32_t words, hipStream_t *out)
{
    device_pool *p = pool_of(device);
    if (p) {
        std::lock_guard<std::timed_mutex> lock(p->mu);
        for (size_t i = 0; i < p->streams.size(); ++i)
            if (p->streams[i].first == first_group && p->streams[i].last == last_group) {
                *out = p->streams[i].stream;
                p->streams.erase(p->streams.begin() + (long)i);
                return hipSuccess;
            }
    }
    if (first_group < 0)
        return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
    return hipExtStreamCreateWithCUMask(out, words, mask);
}

void skm_stream_release(int device, int first_group, int last_group, hipStream_t s)
{
    if (!s)
        return;
    device_pool *p = pool_of(device);
    if (!p) {
        (void)hipStreamDestroy(s);
        return;
    }
    if (hipStreamSynchronize(s) != hipSuccess)
        (void)hipGetLastError();
    std::lock_guard<std::timed_mutex> lock(p->mu);
    p->streams.push_back({first_group, last_group, s});
}

hipEvent_t skm_event_acquire(int device, bool timing)
{
    device_pool *p = pool_of(device);
    if (!p)
        return nullptr;
    std::lock_guard<std::timed_mutex> lock(p->mu);
    return take_event(p, timing ? 1 : 0);
}

void skm_event_release(int device, bool timing, hipEvent_t e)
{
    if (!e)
        return;
    device_pool *p = pool_of(device);
    if (!p) {
        (void)hipEventDestroy(e);
        return;
    }
    std::lock_guard<std::timed_mutex> lock(p->mu);
    p->events[timing ? 1 : 0].push_back(e);
}

hipError_t skm_pinned_acquire(int device, void **out)
{
    device_pool *p = pool_of(device);
    if (p) {
        std::lock_guard<std::timed_mutex> lock(p->mu);
        if (!p->pinned_pages.empty()) {
            *out = p->pinned_pages.back();
            p->pinned_pages.pop_back();
            memset(*out, 0, 4096);
            return hipSuccess;
        }
    }
    const hipError_t e = hipHostMalloc(out, 4096, hipHostMallocDefault);
    if (e == hipSuccess)
        memset(*out, 0, 4096);
    return e;
}

// the caller has made sure that no kernel still writes the page (skm_destroy waits for every stream of the device)
void skm_pinned_release(int device, void *page)
{
    if (!page)
        return;
    device_pool *p = pool_of(device);
    if (!p) {
        (void)hipHostFree(page);
        return;
    }
    std::lock_guard<std::timed_mutex> lock(p->mu);
    p->pinned_pages.push_back(page);
}

// ---------------------------------------------------------------------------- device memory
int skm_pool_alloc(skm_ctx *ctx, size_t bytes, void **out)
{
    *out = nullptr;
    device_pool *p = pool_of(ctx->device);
    if (!p) {  // the process is exiting (or an impossible device number)
        SKM_HIP(hipMalloc(out, bytes ? bytes : 1));
        return SKM_OK;
    }
    const bool guard = guard_on();
    const size_t user = bytes ? bytes : 1;
    const size_t cls = size_class(user + (guard ? GUARD_BYTES : 0));
    void *ptr = nullptr;
    {
        std::lock_guard<std::timed_mutex> lock(p->mu);
        drain_limbo_locked(p);
        auto range = p->parked.equal_range(cls);
        int looked = 0;
        for (auto it = range.first; it != range.second && looked < 8; ++it, ++looked) {
            if (!block_idle(p, it->second)) {
                ++p->n_wait_skipped;
                continue;
            }
            ptr = it->second.ptr;
            p->parked.erase(it);
            p->parked_bytes -= (int64_t)cls;
            ++p->n_reuse;
            break;
        }
        if (!ptr) {
            hipError_t e = hipMalloc(&ptr, cls);
            if (e != hipSuccess) {  // out of memory: everything parked goes back to the runtime, then once more
                (void)hipGetLastError();
                ptr = nullptr;
                trim_locked(p);
                e = hipMalloc(&ptr, cls);
                if (e != hipSuccess) {
                    (void)hipGetLastError();
                    skm_set_error("hipMalloc(%zu bytes for a request of %zu): %s (%lld bytes live in this library's arrays)", cls, bytes,
                                  hipGetErrorString(e), (long long)p->live_bytes);
                    return SKM_E_NOMEM;
                }
            }
            ++p->n_malloc;
        }
        live_block lb;
        lb.class_bytes = cls;
        lb.user_bytes = user;
        lb.guarded = guard;
        p->live[ptr] = lb;
        p->live_bytes += (int64_t)cls;
    }
    if (guard && !ctx->capturing)
        SKM_HIP(hipMemsetAsync((uint8_t *)ptr + user, GUARD_BYTE, GUARD_BYTES, ctx->stream));
    *out = ptr;
    return SKM_OK;
}

int skm_pool_free(skm_ctx *ctx, void *ptr)
{
    if (!ptr)
        return SKM_OK;
    device_pool *p = pool_of(ctx->device);
    if (!p) {  // the process is exiting: the pool has been handed back
        (void)hipDeviceSynchronize();
        (void)hipFree(ptr);
        (void)hipGetLastError();
        return SKM_OK;
    }
    live_block lb;
    {
        std::lock_guard<std::timed_mutex> lock(p->mu);
        auto it = p->live.find(ptr);
        if (it == p->live.end()) {
            // not one of ours (the header allows "any hipMalloc"): the runtime's own free, behind an explicit wait
            quiesce_locked(p);
            SKM_HIP(hipFree(ptr));
            return SKM_OK;
        }
        lb = it->second;
    }
    int rc = SKM_OK;
    if (lb.guarded && !ctx->capturing) {
        // (outside the lock: a host wait)  every stream that may have written the array must be done before the look
        unsigned char host[GUARD_BYTES];
        SKM_TRY(skm_quiesce_device(ctx->device));
        SKM_HIP(hipMemcpyAsync(host, (uint8_t *)ptr + lb.user_bytes, GUARD_BYTES, hipMemcpyDeviceToHost, ctx->stream));
        SKM_HIP(hipStreamSynchronize(ctx->stream));
        for (size_t i = 0; i < GUARD_BYTES; ++i)
            if (host[i] != GUARD_BYTE) {
                skm_set_error("SKM_GUARD: an array of %zu bytes was overrun: byte %zu behind its end holds 0x%02x", lb.user_bytes, i,
                              (unsigned)host[i]);
                rc = SKM_E_HIP;
                break;
            }
    }
    std::lock_guard<std::timed_mutex> lock(p->mu);
    p->live.erase(ptr);
    p->live_bytes -= (int64_t)lb.class_bytes;
    if (any_capture(p)) {
        p->limbo.emplace_back(ptr, lb.class_bytes);
    } else {
        drain_limbo_locked(p);
        park_locked(p, ptr, lb.class_bytes);
    }
    return rc;
}

extern "C" int skm_mem_trim(skm_ctx *ctx, int64_t *h_released_bytes)
{
    SKM_REQUIRE(ctx, SKM_E_BADARG, "null context");
    SKM_HIP(hipSetDevice(ctx->device));
    device_pool *p = pool_of(ctx->device);
    SKM_REQUIRE(p, SKM_E_BADARG, "skm_mem_trim: device %d out of range", ctx->device);
    std::lock_guard<std::timed_mutex> lock(p->mu);
    drain_limbo_locked(p);
    const int64_t released = trim_locked(p);
    if (h_released_bytes)
        *h_released_bytes = released;
    return SKM_OK;
}

extern "C" int skm_mem_stats(skm_ctx *ctx, int64_t *h_out8)
{
    SKM_REQUIRE(ctx && h_out8, SKM_E_BADARG, "skm_mem_stats: null argument");
    device_pool *p = pool_of(ctx->device);
    SKM_REQUIRE(p, SKM_E_BADARG, "skm_mem_stats: device %d out of range", ctx->device);
    std::lock_guard<std::timed_mutex> lock(p->mu);
    h_out8[0] = p->live_bytes;
    h_out8[1] = p->parked_bytes;
    h_out8[2] = p->n_malloc;
    h_out8[3] = p->n_free;
    h_out8[4] = p->n_reuse;
    h_out8[5] = (int64_t)p->parked.size() + (int64_t)p->limbo.size();
    h_out8[6] = (int64_t)p->streams.size();
    h_out8[7] = p->n_wait_skipped;
    return SKM_OK;
}

// ---------------------------------------------------------------------------- diagnostics
// For a watchdog thread while another thread does not come back from a call: per context, whether its streams are idle,
// the first timed kernel (skm_profile_enable) whose stop event has not completed, and what the pool holds.  Takes the pool
// lock for at most a quarter of a second (the stuck thread may hold it) and reads without it otherwise.
extern "C" int skm_debug_report(char *h_buf, int cap)
{
    SKM_REQUIRE(h_buf && cap > 0, SKM_E_BADARG, "skm_debug_report: bad argument");
    std::string out;
    char line[512];
    for (int d = 0; d < MAX_DEVICES; ++d) {
        device_pool *p;
        {
            std::lock_guard<std::mutex> lock(g_pools_mu);
            p = g_pools[d];
        }
        if (!p)
            continue;
        const bool locked = p->mu.try_lock_for(std::chrono::milliseconds(250));
        snprintf(line, sizeof(line),
                 "device %d%s: %zu contexts; arrays live %lld B, parked %lld B in %zu blocks (+%zu in limbo); hipMalloc %lld, hipFree %lld, reused %lld, "
                 "trims %lld; cached streams %zu\n",
                 d, locked ? "" : " (pool lock HELD by another thread)", p->contexts.size(), (long long)p->live_bytes,
                 (long long)p->parked_bytes, p->parked.size(), p->limbo.size(), (long long)p->n_malloc, (long long)p->n_free,
                 (long long)p->n_reuse, (long long)p->n_trim, p->streams.size());
        out += line;
        (void)hipSetDevice(d);
        for (const skm_ctx *c : p->contexts) {
            auto state = [](hipStream_t s) -> const char * {
                if (!s)
                    return "-";
                const hipError_t e = hipStreamQuery(s);
                if (e != hipSuccess)
                    (void)hipGetLastError();
                return e == hipSuccess ? "idle" : (e == hipErrorNotReady ? "BUSY" : hipGetErrorString(e));
            };
            snprintf(line, sizeof(line), "  ctx %p (CUs %d of %d)%s: stream %s, writer stream %s, gram stream %s; %zu timed kernels recorded\n",
                     (const void *)c, c->usable_cus, c->num_cus, c->capturing ? " CAPTURING" : "", c->capturing ? "?" : state(c->stream),
                     state(c->s_writer), state(c->s_gram), c->prof.size());
            out += line;
            size_t shown = 0;
            for (size_t i = 0; i < c->prof.size() && shown < 6; ++i) {
                const hipError_t e = hipEventQuery(c->prof[i].stop);
                if (e == hipSuccess)
                    continue;
                (void)hipGetLastError();
                const bool started = hipEventQuery(c->prof[i].start) == hipSuccess;
                (void)hipGetLastError();
                snprintf(line, sizeof(line), "    timed kernel %zu of %zu `%s`: %s\n", i, c->prof.size(), c->prof[i].name,
                         started ? "STARTED, NOT FINISHED" : "not started");
                out += line;
                ++shown;
            }
        }
        if (locked)
            p->mu.unlock();
    }
    if (out.empty())
        out = "no context was ever created\n";
    strncpy(h_buf, out.c_str(), (size_t)cap - 1);
    h_buf[cap - 1] = 0;
    return SKM_OK;
}
