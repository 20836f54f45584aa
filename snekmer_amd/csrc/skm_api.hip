// Context, memory, error and per-kernel timing plumbing of libsnekmer_hip.so.
#include <algorithm>
#include <cstring>
#include <map>

#include <cstdlib>

#include "skm_common.h"

extern "C" int skm_abi_version(void) { return SKM_ABI_VERSION; }

// ---- options: the environment is read once; skm_set_option changes a value afterwards (include/snekmer_hip.h)
namespace {
struct option_desc {
    const char *name;
    int skm_options::*field;
    int unset;                 // the field's value when the option is not set
    const char *const *words;  // named values (their index + 1 is the field's value), or nullptr: an integer in [lo, hi]
    int lo, hi;
    bool diag;                 // only read by -DSKM_DIAG builds
};
const char *const SORT_WORDS[] = {"rocprim", "onesweep", nullptr};
const char *const PATH_WORDS[] = {"lists", "cursor", nullptr};
const option_desc OPTIONS[] = {
    {"SKM_SORT", &skm_options::sort, 0, SORT_WORDS, 0, 0, false},
    {"SKM_COSINE_PATH", &skm_options::cosine_path, 0, PATH_WORDS, 0, 0, false},
    {"SKM_HEAVY_PANEL", &skm_options::heavy_panel, -1, nullptr, 0, 1, false},
    {"SKM_HEAVY_PACK", &skm_options::heavy_pack, -1, nullptr, 0, 1, false},
    {"SKM_COSINE_OVERLAP", &skm_options::cosine_overlap, 0, nullptr, 0, 1, false},
    {"SKM_GRAM_SHAPE", &skm_options::gram_shape, 0, nullptr, 0, 4, false},
    {"SKM_DENSE_VARIANT", &skm_options::dense_variant, 0, nullptr, 0, 11, false},
#ifdef SKM_DIAG  // (the product library does not even carry the names: tests/test_host_api.py)
    {"SKM_COSINE_ABLATE", &skm_options::cosine_ablate, 0, nullptr, 0, 3, true},
    {"SKM_GRAM_ABLATE", &skm_options::gram_ablate, 0, nullptr, 0, 4, true},
    {"SKM_OVERLAP_BLOCKS", &skm_options::overlap_blocks, 0, nullptr, 0, 16, true},
    {"SKM_DENSE_SPLIT", &skm_options::dense_split, 0, nullptr, 0, 8, true},
    {"SKM_HEAVY_ABLATE", &skm_options::heavy_ablate, 0, nullptr, 0, 31, true},
#endif
};
skm_options g_env_opts, g_opts;
bool g_opts_read = false;

bool parse_option(const option_desc &d, const char *value, int *out)
{
    if (!value || !*value) {
        *out = d.unset;
        return true;
    }
    if (d.words) {
        for (int i = 0; d.words[i]; ++i)
            if (strcmp(value, d.words[i]) == 0) {
                *out = i + 1;
                return true;
            }
        return false;
    }
    char *end = nullptr;
    const long v = strtol(value, &end, 10);
    if (end == value || *end || v < d.lo || v > d.hi)
        return false;
    *out = (int)v;
    return true;
}

void read_options_once()
{
    if (g_opts_read)
        return;
    for (const option_desc &d : OPTIONS) {
        int v = d.unset;
        const char *e = getenv(d.name);
        if (!parse_option(d, e, &v)) {
            fprintf(stderr, "libsnekmer_hip: %s=%s is not a value of that option: ignored\n", d.name, e);
            v = d.unset;
        }
        g_env_opts.*(d.field) = v;
    }
    g_opts = g_env_opts;
    g_opts_read = true;
}
}  // namespace

const skm_options &skm_opts()
{
    read_options_once();
    return g_opts;
}

extern "C" int skm_set_option(const char *name, const char *value)
{
    SKM_REQUIRE(name, SKM_E_BADARG, "skm_set_option: null name");
    read_options_once();
    for (const option_desc &d : OPTIONS)
        if (strcmp(name, d.name) == 0) {
            int v = g_env_opts.*(d.field);
            if (value)
                SKM_REQUIRE(parse_option(d, value, &v), SKM_E_BADARG, "skm_set_option: %s does not take the value \"%s\"", name, value);
            g_opts.*(d.field) = v;
            return SKM_OK;
        }
    skm_set_error("skm_set_option: no option named %s", name);
    return SKM_E_BADARG;
}

extern "C" int skm_get_option(const char *name, char *buf, int cap)
{
    SKM_REQUIRE(name && buf && cap > 0, SKM_E_BADARG, "skm_get_option: bad argument");
    read_options_once();
    for (const option_desc &d : OPTIONS)
        if (strcmp(name, d.name) == 0) {
            const int v = g_opts.*(d.field);
            if (v == d.unset)
                buf[0] = 0;
            else if (d.words)
                snprintf(buf, (size_t)cap, "%s", d.words[v - 1]);
            else
                snprintf(buf, (size_t)cap, "%d", v);
            return SKM_OK;
        }
    skm_set_error("skm_get_option: no option named %s", name);
    return SKM_E_BADARG;
}

extern "C" int skm_device_count(int *h_count)
{
    SKM_REQUIRE(h_count, SKM_E_BADARG, "skm_device_count: null output");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *h_count = 0;
        skm_set_error("hipGetDeviceCount: %s", hipGetErrorString(e));
        return SKM_E_HIP;
    }
    *h_count = n;
    return SKM_OK;
}

namespace {
// first_group < 0: an ordinary stream.  Otherwise the context's stream is confined to the CU groups first..last of 0..7
// (mask bit i: group (i / 8) % 8 - every group has eight CUs in every XCD -, as the two streams of the overlapped cosine
// schedule are).
int create_ctx(int device_id, int first_group, int last_group, skm_ctx **out_ctx, const char *who)
{
    SKM_REQUIRE(out_ctx, SKM_E_BADARG, "%s: null output", who);
    *out_ctx = nullptr;
    int n = 0;
    SKM_HIP(hipGetDeviceCount(&n));
    SKM_REQUIRE(device_id >= 0 && device_id < n, SKM_E_BADARG, "%s: device %d out of range (%d visible)", who, device_id, n);
    SKM_HIP(hipSetDevice(device_id));
    hipDeviceProp_t prop;
    SKM_HIP(hipGetDeviceProperties(&prop, device_id));
    hipStream_t stream = nullptr;
    int usable = prop.multiProcessorCount;
    if (first_group >= 0) {
        const int ncu = prop.multiProcessorCount;
        SKM_REQUIRE(first_group <= last_group && last_group <= 7, SKM_E_BADARG, "%s: CU groups %d..%d (want 0 <= first <= last <= 7)", who,
                    first_group, last_group);
        // The mask layout below (bit i = compute unit i, group (i / 8) % 8, every group holding eight CUs of every XCD) was
        // verified on MI355X only: 256 CUs in 8 XCDs.  Anything else gets no confined stream (the callers fall back to an
        // ordinary context).  NOTE: hipExtStreamCreateWithCUMask has no flags argument - the stream it returns is a BLOCKING
        // stream, i.e. it synchronises implicitly with the legacy null stream of the process.  The library itself never
        // uses the null stream (every copy and launch names a stream); a host program that does (e.g. another framework's
        // default stream in the same process) serialises against confined contexts.
        SKM_REQUIRE(strncmp(prop.gcnArchName, "gfx950", 6) == 0 && ncu == 256, SKM_E_UNSUPPORTED,
                    "%s: CU-group masks are defined for gfx950 with 256 compute units (this device: %s, %d)", who, prop.gcnArchName, ncu);
        uint32_t mask[16] = {};
        for (int i = 0; i < ncu; ++i)
            if ((i / 8) % 8 >= first_group && (i / 8) % 8 <= last_group)
                mask[i / 32] |= 1u << (i % 32);
        usable = 0;
        for (int i = 0; i < ncu; ++i)
            usable += (mask[i / 32] >> (i % 32)) & 1u;
        // (a stream of the same mask that an earlier context gave back is taken again: skm_mem.hip)
        SKM_HIP(skm_stream_acquire(device_id, first_group, last_group, mask, (uint32_t)((ncu + 31) / 32), &stream));
    } else {
        SKM_HIP(skm_stream_acquire(device_id, -1, -1, nullptr, 0, &stream));
    }
    skm_ctx *ctx = new skm_ctx();
    ctx->device = device_id;
    ctx->num_cus = prop.multiProcessorCount;
    ctx->usable_cus = usable;
    ctx->stream = stream;
    ctx->cu_first = first_group;
    ctx->cu_last = last_group;
    {
        const hipError_t e = skm_pinned_acquire(device_id, &ctx->h_pinned);  // zeroed (offset 2048: the previous cosine call's heavy-row count)
        if (e != hipSuccess) {
            (void)hipGetLastError();
            skm_stream_release(device_id, first_group, last_group, stream);
            delete ctx;
            skm_set_error("%s: pinned page: %s", who, hipGetErrorString(e));
            return SKM_E_NOMEM;
        }
    }
    void *dev = nullptr;
    if (hipHostGetDevicePointer(&dev, ctx->h_pinned, 0) == hipSuccess && dev)
        ctx->d_err = (uint32_t *)dev + SKM_DEVERR_WORD;
    else
        (void)hipGetLastError();
    skm_registry_add(ctx);
    *out_ctx = ctx;
    return SKM_OK;
}
}  // namespace

extern "C" int skm_create(int device_id, skm_ctx **out_ctx) { return create_ctx(device_id, -1, -1, out_ctx, "skm_create"); }

extern "C" int skm_create_confined(int device_id, int first_cu_group, int last_cu_group, skm_ctx **out_ctx)
{
    SKM_REQUIRE(first_cu_group >= 0, SKM_E_BADARG, "skm_create_confined: CU groups %d..%d (want 0 <= first <= last <= 7)", first_cu_group,
                last_cu_group);
    return create_ctx(device_id, first_cu_group, last_cu_group, out_ctx, "skm_create_confined");
}

extern "C" int skm_destroy(skm_ctx *ctx)
{
    if (!ctx)
        return SKM_OK;
    hipSetDevice(ctx->device);
    if (ctx->capturing) {  // an abandoned capture: end it so that the stream can be waited for and reused
        hipGraph_t g = nullptr;
        if (hipStreamEndCapture(ctx->stream, &g) == hipSuccess && g)
            (void)hipGraphDestroy(g);
        (void)hipGetLastError();
        ctx->capturing = false;
    }
    // Every stream of the device, not only this context's: another context's stream may be running a kernel that reads this
    // context's scratch or writes its pinned words (skm_cosine_csr_phase 2 runs on the executing context's stream).  An
    // explicit wait - nothing below relies on the runtime waiting inside a free.
    skm_quiesce_device(ctx->device);
    skm_comm_destroy(ctx);
    for (auto g : ctx->graphs)  // graphs the caller never destroyed (their handles are dead from here on)
        skm_graph_release(g);
    ctx->graphs.clear();
    for (auto &p : ctx->prof) {
        skm_event_release(ctx->device, true, p.start);
        skm_event_release(ctx->device, true, p.stop);
    }
    for (auto e : ctx->event_pool)
        skm_event_release(ctx->device, true, e);
    skm_registry_remove(ctx);
    for (int i = 0; i < WS_COUNT; ++i)
        if (ctx->ws[i])
            skm_pool_free(ctx, ctx->ws[i]);
    skm_pinned_release(ctx->device, ctx->h_pinned);
    skm_event_release(ctx->device, false, ctx->ev_host);
    for (auto e : ctx->sync_events)
        skm_event_release(ctx->device, false, e);
    for (auto e : ctx->user_events)
        skm_event_release(ctx->device, false, e);
    skm_stream_release(ctx->device, -1, -1, ctx->s_writer);
    skm_stream_release(ctx->device, 0, 3, ctx->s_gram);
    skm_stream_release(ctx->device, ctx->cu_first, ctx->cu_last, ctx->stream);
    delete ctx;
    return SKM_OK;
}

// Ordering between two contexts of one process (each context is one stream): `skm_event_record` marks the current end
// of this context's stream in one of its SKM_EVENT_SLOTS slots; `skm_stream_wait` makes everything queued on `ctx`
// from now on wait for the mark last recorded in `src`'s slot.  Neither blocks the host.
extern "C" int skm_event_record(skm_ctx *ctx, int slot)
{
    SKM_REQUIRE(ctx && slot >= 0 && slot < SKM_EVENT_SLOTS, SKM_E_BADARG, "skm_event_record: bad argument");
    SKM_HIP(hipSetDevice(ctx->device));
    if (ctx->user_events.empty())
        ctx->user_events.assign(SKM_EVENT_SLOTS, nullptr);
    if (!ctx->user_events[slot]) {
        ctx->user_events[slot] = skm_event_acquire(ctx->device, false);
        SKM_REQUIRE(ctx->user_events[slot], SKM_E_HIP, "skm_event_record: no event");
    }
    SKM_HIP(hipEventRecord(ctx->user_events[slot], ctx->stream));
    return SKM_OK;
}

extern "C" int skm_event_query(skm_ctx *ctx, int slot, int *h_done)
{
    SKM_REQUIRE(ctx && h_done && slot >= 0 && slot < SKM_EVENT_SLOTS, SKM_E_BADARG, "skm_event_query: bad argument");
    *h_done = 1;
    if (ctx->user_events.empty() || !ctx->user_events[slot])
        return SKM_OK;
    const hipError_t e = hipEventQuery(ctx->user_events[slot]);
    if (e == hipErrorNotReady) {
        (void)hipGetLastError();
        *h_done = 0;
        return SKM_OK;
    }
    SKM_HIP(e);
    return SKM_OK;
}

extern "C" int skm_stream_wait(skm_ctx *ctx, skm_ctx *src, int slot)
{
    SKM_REQUIRE(ctx && src && slot >= 0 && slot < SKM_EVENT_SLOTS, SKM_E_BADARG, "skm_stream_wait: bad argument");
    SKM_REQUIRE(ctx->device == src->device, SKM_E_BADARG, "skm_stream_wait: contexts on different devices");
    if (src->user_events.empty() || !src->user_events[slot])
        return SKM_OK;  // nothing was ever recorded there: nothing to wait for
    SKM_HIP(hipSetDevice(ctx->device));
    SKM_HIP(hipStreamWaitEvent(ctx->stream, src->user_events[slot], 0));
    return SKM_OK;
}

int skm_check_device_error(skm_ctx *ctx, const char *who)
{
    volatile uint32_t *w = (volatile uint32_t *)ctx->h_pinned + SKM_DEVERR_WORD;
    const uint32_t bits = *w;
    if (!bits)
        return SKM_OK;
    *w = 0;
    if (bits & SKM_DEVERR_SEQ_TOO_LONG) {
        skm_set_error("%s: an earlier skm_count_csr / skm_vectorize_csr call on this context was given a max_seq_len smaller than "
                      "one of its sequences; the rows of those sequences were left EMPTY (pass 0 or a true bound)", who);
        return SKM_E_BADARG;
    }
    skm_set_error("%s: a kernel reported error bits 0x%x", who, bits);
    return SKM_E_HIP;
}

extern "C" int skm_sync(skm_ctx *ctx)
{
    SKM_REQUIRE(ctx, SKM_E_BADARG, "null context");
    SKM_HIP(hipStreamSynchronize(ctx->stream));
    return skm_check_device_error(ctx, "skm_sync");
}

extern "C" int skm_device_info(skm_ctx *ctx, char *h_name, int name_cap, int *h_cus, int64_t *h_mem_bytes)
{
    SKM_REQUIRE(ctx, SKM_E_BADARG, "null context");
    hipDeviceProp_t prop;
    SKM_HIP(hipGetDeviceProperties(&prop, ctx->device));
    if (h_name && name_cap > 0) {
        snprintf(h_name, (size_t)name_cap, "%s (%s)", prop.name, prop.gcnArchName);
    }
    if (h_cus)
        *h_cus = prop.multiProcessorCount;
    if (h_mem_bytes)
        *h_mem_bytes = (int64_t)prop.totalGlobalMem;
    return SKM_OK;
}

// ---------------------------------------------------------------------------- memory
extern "C" int skm_malloc(skm_ctx *ctx, size_t bytes, void **out_dptr)
{
    SKM_REQUIRE(ctx && out_dptr, SKM_E_BADARG, "skm_malloc: null argument");
    *out_dptr = nullptr;
    SKM_HIP(hipSetDevice(ctx->device));
    return skm_pool_alloc(ctx, bytes, out_dptr);  // (skm_mem.hip: size classes, reuse behind completed events only)
}

// Never waits and never calls hipFree: the block is parked behind an event on every stream of the device that still has
// work queued, and handed out again only once those have completed.  Letting an array go while kernels that read it are
// queued is therefore safe by construction (it used to rest on hipFree's implicit device-wide wait).
extern "C" int skm_free(skm_ctx *ctx, void *dptr)
{
    SKM_REQUIRE(ctx, SKM_E_BADARG, "null context");
    if (!dptr)
        return SKM_OK;
    SKM_HIP(hipSetDevice(ctx->device));
    return skm_pool_free(ctx, dptr);
}

extern "C" int skm_host_alloc(skm_ctx *ctx, size_t bytes, void **out_hptr)
{
    SKM_REQUIRE(ctx && out_hptr, SKM_E_BADARG, "skm_host_alloc: null argument");
    *out_hptr = nullptr;
    SKM_HIP(hipSetDevice(ctx->device));
    hipError_t e = hipHostMalloc(out_hptr, bytes ? bytes : 1, hipHostMallocMapped | hipHostMallocCoherent);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        skm_set_error("hipHostMalloc(%zu bytes): %s", bytes, hipGetErrorString(e));
        return SKM_E_NOMEM;
    }
    void *dev = nullptr;
    e = hipHostGetDevicePointer(&dev, *out_hptr, 0);
    if (e != hipSuccess || dev != *out_hptr) {  // the callers rely on one address for both sides
        (void)hipGetLastError();
        (void)hipHostFree(*out_hptr);
        *out_hptr = nullptr;
        skm_set_error("skm_host_alloc: pinned memory is not device-addressable at its host address");
        return SKM_E_UNSUPPORTED;
    }
    return SKM_OK;
}

extern "C" int skm_host_free(skm_ctx *ctx, void *hptr)
{
    SKM_REQUIRE(ctx, SKM_E_BADARG, "null context");
    if (hptr) {
        SKM_HIP(hipSetDevice(ctx->device));
        SKM_TRY(skm_quiesce_device(ctx->device));  // kernels read and write this memory in place; the wait is explicit
        SKM_HIP(hipHostFree(hptr));
    }
    return SKM_OK;
}

extern "C" int skm_memcpy_h2d(skm_ctx *ctx, void *d_dst, const void *h_src, size_t bytes)
{
    SKM_REQUIRE(ctx, SKM_E_BADARG, "null context");
    if (!bytes)
        return SKM_OK;
    SKM_HIP(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream));
    SKM_HIP(hipStreamSynchronize(ctx->stream));
    return SKM_OK;
}

extern "C" int skm_memcpy_h2d_async(skm_ctx *ctx, void *d_dst, const void *h_src_pinned, size_t bytes)
{
    SKM_REQUIRE(ctx, SKM_E_BADARG, "null context");
    if (!bytes)
        return SKM_OK;
    SKM_REQUIRE(d_dst && h_src_pinned, SKM_E_BADARG, "skm_memcpy_h2d_async: null array");
    SKM_HIP(hipMemcpyAsync(d_dst, h_src_pinned, bytes, hipMemcpyHostToDevice, ctx->stream));
    return SKM_OK;
}

extern "C" int skm_memcpy_d2h(skm_ctx *ctx, void *h_dst, const void *d_src, size_t bytes)
{
    SKM_REQUIRE(ctx, SKM_E_BADARG, "null context");
    if (!bytes)
        return SKM_OK;
    SKM_HIP(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    SKM_HIP(hipStreamSynchronize(ctx->stream));
    return skm_check_device_error(ctx, "skm_memcpy_d2h");
}

extern "C" int skm_memcpy_d2d(skm_ctx *ctx, void *d_dst, const void *d_src, size_t bytes)
{
    SKM_REQUIRE(ctx, SKM_E_BADARG, "null context");
    if (!bytes)
        return SKM_OK;
    SKM_HIP(hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return SKM_OK;
}

extern "C" int skm_memset(skm_ctx *ctx, void *d_dst, int byte_value, size_t bytes)
{
    SKM_REQUIRE(ctx, SKM_E_BADARG, "null context");
    if (!bytes)
        return SKM_OK;
    SKM_HIP(hipMemsetAsync(d_dst, byte_value, bytes, ctx->stream));
    return SKM_OK;
}

int skm_ws(skm_ctx *ctx, int slot, size_t bytes, void **out)
{
    if (bytes < 256)
        bytes = 256;
    if (ctx->ws_bytes[slot] < bytes) {
        if (ctx->capturing) {
            skm_set_error("scratch slot %d would grow (%zu -> %zu bytes) inside skm_graph_begin .. skm_graph_end: run the same calls "
                          "once outside a capture first", slot, ctx->ws_bytes[slot], bytes);
            return SKM_E_UNSUPPORTED;
        }
        ++ctx->ws_generation;
        if (ctx->ws[slot]) {
            // the old block is parked behind whatever is still queued on any stream (skm_mem.hip): no wait here
            SKM_TRY(skm_pool_free(ctx, ctx->ws[slot]));
            ctx->ws[slot] = nullptr;
            ctx->ws_bytes[slot] = 0;
        }
        size_t want = bytes + bytes / 8;  // headroom so slightly larger batches do not realloc
        {
            const int rc = skm_pool_alloc(ctx, want, &ctx->ws[slot]);
            if (rc != SKM_OK) {
                ctx->ws[slot] = nullptr;
                return rc;
            }
        }
        ctx->ws_bytes[slot] = want;
        if (slot == WS_COS || slot == WS_ZERO)  // hold counters that every user leaves at zero (k_cosine_prologue, k_compact_rows)
            SKM_HIP(hipMemsetAsync(ctx->ws[slot], 0, want, ctx->stream));
    }
    *out = ctx->ws[slot];
    return SKM_OK;
}

// ---------------------------------------------------------------------------- graphs
struct skm_graph {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    uint64_t ws_generation = 0;
    skm_ctx *owner = nullptr;
    size_t nodes = 0;  // kernels, fills and copies the capture holds: the GPU operations of one replay
};

void skm_graph_release(skm_graph *gr)
{
    if (!gr)
        return;
    if (gr->exec)
        (void)hipGraphExecDestroy(gr->exec);
    if (gr->graph)
        (void)hipGraphDestroy(gr->graph);
    delete gr;
}

extern "C" int skm_graph_begin(skm_ctx *ctx)
{
    SKM_REQUIRE(ctx, SKM_E_BADARG, "null context");
    SKM_REQUIRE(!ctx->capturing, SKM_E_BADARG, "skm_graph_begin: a capture is already open on this context");
    SKM_HIP(hipSetDevice(ctx->device));
    // relaxed mode: other threads (and this one) may allocate and free while the capture is open; what is captured is
    // exactly what the library queues on this context's stream
    SKM_HIP(hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeRelaxed));
    ctx->capturing = true;
    return SKM_OK;
}

extern "C" int skm_graph_end(skm_ctx *ctx, skm_graph **out_graph)
{
    SKM_REQUIRE(ctx && out_graph, SKM_E_BADARG, "skm_graph_end: null argument");
    *out_graph = nullptr;
    SKM_REQUIRE(ctx->capturing, SKM_E_BADARG, "skm_graph_end: no capture is open on this context");
    ctx->capturing = false;
    hipGraph_t g = nullptr;
    hipError_t e = hipStreamEndCapture(ctx->stream, &g);
    if (e != hipSuccess || !g) {
        (void)hipGetLastError();
        skm_set_error("skm_graph_end: the capture was invalidated (%s): a call inside it waited for the device or used another stream",
                      hipGetErrorString(e));
        return SKM_E_HIP;
    }
    hipGraphExec_t x = nullptr;
    e = hipGraphInstantiate(&x, g, nullptr, nullptr, 0);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        (void)hipGraphDestroy(g);
        skm_set_error("hipGraphInstantiate: %s", hipGetErrorString(e));
        return SKM_E_HIP;
    }
    skm_graph *gr = new skm_graph();
    gr->graph = g;
    gr->exec = x;
    gr->ws_generation = ctx->ws_generation;
    gr->owner = ctx;
    (void)hipGraphGetNodes(g, nullptr, &gr->nodes);
    ctx->graphs.push_back(gr);
    *out_graph = gr;
    return SKM_OK;
}

extern "C" int skm_graph_nodes(skm_graph *graph, int64_t *h_nodes)
{
    SKM_REQUIRE(graph && h_nodes, SKM_E_BADARG, "skm_graph_nodes: null argument");
    *h_nodes = (int64_t)graph->nodes;
    return SKM_OK;
}

extern "C" int skm_graph_launch(skm_ctx *ctx, skm_graph *graph)
{
    SKM_REQUIRE(ctx && graph && graph->owner == ctx, SKM_E_BADARG, "skm_graph_launch: the graph was captured on another context");
    SKM_REQUIRE(!ctx->capturing, SKM_E_BADARG, "skm_graph_launch: a capture is open on this context");
    if (graph->ws_generation != ctx->ws_generation) {
        skm_set_error("skm_graph_launch: a scratch buffer of the context was reallocated since the capture; capture again");
        return SKM_E_STALE;
    }
    SKM_HIP(hipSetDevice(ctx->device));
    SKM_HIP(hipGraphLaunch(graph->exec, ctx->stream));
    return SKM_OK;
}

extern "C" int skm_graph_destroy(skm_ctx *ctx, skm_graph *graph)
{
    SKM_REQUIRE(ctx, SKM_E_BADARG, "null context");
    if (!graph)
        return SKM_OK;
    auto it = std::find(ctx->graphs.begin(), ctx->graphs.end(), graph);
    SKM_REQUIRE(it != ctx->graphs.end(), SKM_E_BADARG, "skm_graph_destroy: not a live graph of this context");
    SKM_HIP(hipSetDevice(ctx->device));
    (void)hipStreamSynchronize(ctx->stream);  // a replay may still be running
    ctx->graphs.erase(it);
    skm_graph_release(graph);
    return SKM_OK;
}

// ---------------------------------------------------------------------------- profiling
skm_prof_scope::skm_prof_scope(skm_ctx *c, const char *name, hipStream_t on) : ctx(c), st(on ? on : c->stream)
{
    if (!ctx->profiling || ctx->capturing)
        return;
    skm_prof_entry ent;
    ent.name = name;
    auto take = [&](hipEvent_t *ev) {
        if (!ctx->event_pool.empty()) {
            *ev = ctx->event_pool.back();
            ctx->event_pool.pop_back();
        } else {
            *ev = skm_event_acquire(ctx->device, true);
        }
    };
    take(&ent.start);
    take(&ent.stop);
    hipEventRecord(ent.start, st);
    stop = ent.stop;
    ctx->prof.push_back(ent);
}

skm_prof_scope::~skm_prof_scope()
{
    if (stop)
        hipEventRecord(stop, st);
}

extern "C" int skm_profile_enable(skm_ctx *ctx, int on)
{
    SKM_REQUIRE(ctx, SKM_E_BADARG, "null context");
    ctx->profiling = on != 0;
    return SKM_OK;
}

extern "C" int skm_profile_reset(skm_ctx *ctx)
{
    SKM_REQUIRE(ctx, SKM_E_BADARG, "null context");
    SKM_HIP(hipStreamSynchronize(ctx->stream));
    for (auto &p : ctx->prof) {
        ctx->event_pool.push_back(p.start);
        ctx->event_pool.push_back(p.stop);
    }
    ctx->prof.clear();
    return SKM_OK;
}

extern "C" int skm_profile_read(skm_ctx *ctx, const char *h_prefix, int64_t *h_launches, double *h_total_ms)
{
    SKM_REQUIRE(ctx && h_prefix, SKM_E_BADARG, "null argument");
    SKM_HIP(hipStreamSynchronize(ctx->stream));
    int64_t cnt = 0;
    double tot = 0;
    size_t plen = strlen(h_prefix);
    for (auto &p : ctx->prof) {
        if (strncmp(p.name, h_prefix, plen) != 0)
            continue;
        float ms = 0;
        SKM_HIP(hipEventElapsedTime(&ms, p.start, p.stop));
        tot += ms;
        ++cnt;
    }
    if (h_launches)
        *h_launches = cnt;
    if (h_total_ms)
        *h_total_ms = tot;
    return SKM_OK;
}

extern "C" int skm_profile_dump(skm_ctx *ctx, char *h_buf, int cap, int *h_needed)
{
    SKM_REQUIRE(ctx, SKM_E_BADARG, "null context");
    SKM_HIP(hipStreamSynchronize(ctx->stream));
    std::map<std::string, std::pair<int64_t, double>> agg;
    std::vector<std::string> order;
    for (auto &p : ctx->prof) {
        float ms = 0;
        SKM_HIP(hipEventElapsedTime(&ms, p.start, p.stop));
        auto it = agg.find(p.name);
        if (it == agg.end()) {
            order.push_back(p.name);
            agg[p.name] = {1, ms};
        } else {
            it->second.first++;
            it->second.second += ms;
        }
    }
    std::string out;
    char line[256];
    for (auto &nm : order) {
        snprintf(line, sizeof(line), "%s\t%lld\t%.6f\n", nm.c_str(), (long long)agg[nm].first, agg[nm].second);
        out += line;
    }
    if (h_needed)
        *h_needed = (int)out.size() + 1;
    if (h_buf && cap > 0) {
        strncpy(h_buf, out.c_str(), (size_t)cap - 1);
        h_buf[cap - 1] = 0;
    }
    return SKM_OK;
}

// ---------------------------------------------------------------------------- fused vectorize
#ifndef SKM_HIST_IN_COMPACT_MAX
#define SKM_HIST_IN_COMPACT_MAX (1 << 19)
#endif
extern "C" int skm_vectorize_csr(skm_ctx *ctx, const uint8_t *h_rank, int nsym, int k, int code_bits, const uint8_t *d_seq,
                                 const int64_t *d_off, int64_t n, int64_t total_residues, int64_t max_seq_len, int64_t cap_entries,
                                 int64_t *d_rowptr, void *d_codes, uint32_t *d_counts, void *d_basis, uint32_t *d_colidx,
                                 uint32_t *d_colptr, uint64_t *d_post, float *d_rnorm, uint64_t *d_normsq, int64_t *d_ncols)
{
    const bool counts_only = !d_basis && !d_colidx && !d_colptr && !d_post && !d_ncols;  // no basis stage
    SKM_REQUIRE(ctx && h_rank && d_seq && d_off && d_rowptr && d_codes && d_counts &&
                    (counts_only || (d_basis && d_colidx && d_colptr && d_post && d_ncols)) && n >= 1 && total_residues >= 1,
                SKM_E_BADARG, "skm_vectorize_csr: bad argument (empty batches take skm_count_csr + skm_basis_build)");
    SKM_REQUIRE(code_bits == 32 || code_bits == 64, SKM_E_BADARG, "skm_vectorize_csr: code_bits must be 32 or 64");
    SKM_REQUIRE(cap_entries >= total_residues + 1, SKM_E_BADARG,
                "skm_vectorize_csr: cap_entries (%lld) must be > total residues (%lld)", (long long)cap_entries,
                (long long)total_residues);
    SKM_REQUIRE(total_residues < ((int64_t)1 << 32) - 1 && n < ((int64_t)1 << 32) - 1, SKM_E_OVERFLOW,
                "skm_vectorize_csr: more than 2^32 residues or sequences in one batch; split the batch");
    SKM_HIP(hipSetDevice(ctx->device));
    if (counts_only)
        return skm_count_stage_async(ctx, h_rank, nsym, k, code_bits, d_seq, d_off, n, total_residues, max_seq_len, d_rowptr, d_codes, d_counts,
                                     nullptr, d_rnorm, d_normsq);
    void *p;
    SKM_TRY(skm_ws(ctx, WS_K, sizeof(uint64_t) * (size_t)(total_residues + 1), &p));
    uint64_t *rowcount = (uint64_t *)p;
    int key_bits = 0;
    {
        unsigned __int128 space = 1;
        for (int j = 0; j < k; ++j)
            space *= (unsigned)nsym;
        space -= 1;
        while (space) {
            ++key_bits;
            space >>= 1;
        }
        if (key_bits < 1)
            key_bits = 1;
    }
    // no fill operation between the stages: the count stage's last kernel marks colidx and clears the sort state
    skm_count_extras extras;
    extras.colidx_ff = d_colidx;
    SKM_TRY(skm_basis_sort_state(ctx, total_residues, key_bits, code_bits, &extras.zero, &extras.zero_words, &extras.hist_passes,
                                 &extras.hist_key_bits));
    // ... and, for small batches, counts the sort's digit histograms while it writes the codes (one launch less: 0.148 against
    // 0.154 ms per step at 1 000 sequences; from ~0.5 M entries on the LDS atomics cost k_compact_rows more than the
    // histogram kernel takes: 0.045 against 0.015 ms at 2 M entries)
    extras.hist = extras.zero != nullptr && total_residues <= SKM_HIST_IN_COMPACT_MAX;
    SKM_TRY(skm_count_stage_async(ctx, h_rank, nsym, k, code_bits, d_seq, d_off, n, total_residues, max_seq_len, d_rowptr, d_codes, d_counts,
                                  rowcount, d_rnorm, d_normsq, extras));
    return skm_basis_stage_async(ctx, code_bits, key_bits, total_residues, d_rowptr + n, d_codes, rowcount, d_basis, d_colidx,
                                 d_colptr, d_post, d_ncols, extras);
}
