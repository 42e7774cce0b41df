// context.hip — library/context management and error plumbing of libimpop_hip.so.
#include <stdarg.h>
#include <string.h>

#include "device_utils.h"
#include "internal.h"

namespace impop {

static thread_local char g_err[1024] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

int hip_fail(hipError_t e, const char *what, const char *file, int line) {
    set_error("HIP error %d (%s) in %s at %s:%d", (int)e, hipGetErrorString(e), what, file, line);
    if (e == hipErrorOutOfMemory) return IMPOP_E_NOMEM;
    if (e == hipErrorNoDevice || e == hipErrorInvalidDevice) return IMPOP_E_NODEVICE;
    return IMPOP_E_HIP;
}

int ctx_scratch(impop_ctx *ctx, size_t bytes, void **out) {
    if (bytes > ctx->scratch_bytes) {
        if (ctx->scratch) {
            HIP_TRY(hipStreamSynchronize(ctx->stream));
            HIP_TRY(hipFree(ctx->scratch));
            ctx->scratch = nullptr;
            ctx->scratch_bytes = 0;
        }
        size_t want = bytes + (bytes >> 2) + 4096;
        HIP_TRY(hipMalloc(&ctx->scratch, want));
        ctx->scratch_bytes = want;
    }
    *out = ctx->scratch;
    return IMPOP_OK;
}

int ctx_pinned(impop_ctx *ctx, size_t bytes, void **out) {
    if (bytes > ctx->pinned_bytes) {
        if (ctx->pinned) {
            HIP_TRY(hipStreamSynchronize(ctx->stream));
            HIP_TRY(hipHostFree(ctx->pinned));
            ctx->pinned = nullptr;
            ctx->pinned_bytes = 0;
        }
        const size_t want = bytes + (bytes >> 2) + 4096;
        HIP_TRY(hipHostMalloc(&ctx->pinned, want, hipHostMallocDefault));
        ctx->pinned_bytes = want;
    }
    *out = ctx->pinned;
    return IMPOP_OK;
}

int ctx_aux(impop_ctx *ctx, int slot, size_t bytes, void **out) {
    if (bytes > ctx->aux_bytes[slot]) {
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        if (ctx->d_aux[slot]) HIP_TRY(hipFree(ctx->d_aux[slot]));
        ctx->d_aux[slot] = nullptr;
        ctx->aux_bytes[slot] = 0;
        HIP_TRY(hipMalloc(&ctx->d_aux[slot], bytes + (bytes >> 2)));
        ctx->aux_bytes[slot] = bytes + (bytes >> 2);
    }
    *out = ctx->d_aux[slot];
    return IMPOP_OK;
}

__global__ void tajima_consts_kernel(int64_t n, double *out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        TajConsts c = tajima_consts(n);
        out[0] = c.a1; out[1] = c.a2; out[2] = c.b1; out[3] = c.b2;
        out[4] = c.c1; out[5] = c.c2; out[6] = c.e1; out[7] = c.e2;
    }
}

int ensure_tajima_consts(impop_ctx *ctx, int64_t n) {
    if (!ctx->d_taj) HIP_TRY(hipMalloc(&ctx->d_taj, 8 * sizeof(double)));
    if (ctx->taj_n != n) {
        hipLaunchKernelGGL(tajima_consts_kernel, dim3(1), dim3(64), 0, ctx->stream, n, ctx->d_taj);
        HIP_TRY(hipGetLastError());
        ctx->taj_n = n;
    }
    return IMPOP_OK;
}

}  // namespace impop

using namespace impop;

IMPOP_API int impop_version(void) { return IMPOP_ABI_VERSION; }

IMPOP_API const char *impop_last_error(void) { return g_err; }

IMPOP_API int impop_device_count(int *count) {
    REQUIRE(count, "impop_device_count: count is NULL");
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) {
        *count = 0;
        return hip_fail(e, "hipGetDeviceCount", __FILE__, __LINE__);
    }
    *count = c;
    return IMPOP_OK;
}

IMPOP_API int impop_ctx_create(int device, void *stream, impop_ctx **out) {
    REQUIRE(out, "impop_ctx_create: out is NULL");
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0) {
        set_error("impop_ctx_create: no HIP device available (%s); this engine has no CPU fallback",
                  e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
        return IMPOP_E_NODEVICE;
    }
    REQUIRE(device >= 0 && device < count, "impop_ctx_create: device %d out of range [0,%d)", device, count);
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_error("impop_ctx_create: device %d is %s; libimpop_hip.so is built for gfx950 (MI355X) only", device,
                  prop.gcnArchName);
        return IMPOP_E_NODEVICE;
    }
    impop_ctx *ctx = new impop_ctx();
    ctx->device = device;
    ctx->n_cu = prop.multiProcessorCount;
    snprintf(ctx->arch, sizeof ctx->arch, "%s", prop.gcnArchName);
    if (stream) {
        ctx->stream = (hipStream_t)stream;
        ctx->own_stream = false;
    } else {
        hipError_t se = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
        if (se != hipSuccess) {
            delete ctx;
            return hip_fail(se, "hipStreamCreateWithFlags", __FILE__, __LINE__);
        }
        ctx->own_stream = true;
    }
    // the device error word exists from the start (kernels of several streams may point at it)
    hipError_t ee = hipMalloc((void **)&ctx->d_err, sizeof(uint32_t));
    if (ee == hipSuccess) ee = hipMemset(ctx->d_err, 0, sizeof(uint32_t));
    if (ee != hipSuccess) {
        impop_ctx_destroy(ctx);
        return hip_fail(ee, "hipMalloc(device error word)", __FILE__, __LINE__);
    }
    *out = ctx;
    return IMPOP_OK;
}

IMPOP_API int impop_ctx_destroy(impop_ctx *ctx) {
    if (!ctx) return IMPOP_OK;
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    if (ctx->d_taj) hipFree(ctx->d_taj);
    if (ctx->d_queue) hipFree(ctx->d_queue);
    if (ctx->d_err) hipFree(ctx->d_err);
    for (auto &e : ctx->gram_events) { hipEventDestroy(e.first); hipEventDestroy(e.second); }
    if (ctx->scratch) hipFree(ctx->scratch);
    if (ctx->pinned) hipHostFree(ctx->pinned);
    for (void *a : ctx->d_aux)
        if (a) hipFree(a);
    if (ctx->side) { hipStreamSynchronize(ctx->side); hipStreamDestroy(ctx->side); }
    if (ctx->ev_fork) hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join) hipEventDestroy(ctx->ev_join);
    if (ctx->own_stream) hipStreamDestroy(ctx->stream);
    delete ctx;
    return IMPOP_OK;
}

namespace impop {
int ctx_err_fetch(impop_ctx *ctx) {
    ctx->h_err = 0;
    if (ctx->d_err) HIP_TRY(hipMemcpyAsync(&ctx->h_err, ctx->d_err, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    return IMPOP_OK;
}
int ctx_err_result(impop_ctx *ctx, const char *fn) {
    if (!ctx->h_err) return IMPOP_OK;
    const uint32_t w = ctx->h_err;
    ctx->h_err = 0;
    HIP_TRY(hipMemsetAsync(ctx->d_err, 0, sizeof(uint32_t), ctx->stream));
    set_error("%s: internal device check failed (code 0x%x%s); the results of this call are invalid", fn, w,
              (w & DEV_ERR_GROUPING) ? ": greedy grouping made no progress" : "");
    return IMPOP_E_INTERNAL;
}
}  // namespace impop

__global__ void raise_error_kernel(uint32_t *err, uint32_t bits) { atomicOr(err, bits); }

IMPOP_API int impop_debug_raise_device_error(impop_ctx *ctx, uint32_t bits) {
    REQUIRE(ctx && ctx->d_err, "impop_debug_raise_device_error: ctx is NULL");
    HIP_TRY(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(raise_error_kernel, dim3(1), dim3(1), 0, ctx->stream, ctx->d_err, bits);
    HIP_TRY(hipGetLastError());
    return IMPOP_OK;
}

IMPOP_API int impop_ctx_gram_timing(impop_ctx *ctx, int enable) {
    REQUIRE(ctx, "impop_ctx_gram_timing: ctx is NULL");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    ctx->gram_timing = enable != 0;
    ctx->gram_events_used = 0;
    return IMPOP_OK;
}

IMPOP_API int impop_ctx_gram_elapsed(impop_ctx *ctx, double *total_ms, uint64_t *launches) {
    REQUIRE(ctx, "impop_ctx_gram_elapsed: ctx is NULL");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    double t = 0.0;
    for (size_t i = 0; i < ctx->gram_events_used; ++i) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, ctx->gram_events[i].first, ctx->gram_events[i].second));
        t += (double)ms;
    }
    if (total_ms) *total_ms = t;
    if (launches) *launches = ctx->gram_events_used;
    return IMPOP_OK;
}

IMPOP_API int impop_ctx_synchronize(impop_ctx *ctx) {
    REQUIRE(ctx, "impop_ctx_synchronize: ctx is NULL");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return IMPOP_OK;
}

IMPOP_API int impop_ctx_device_name(impop_ctx *ctx, char *buf, size_t buflen) {
    REQUIRE(ctx && buf && buflen, "impop_ctx_device_name: bad arguments");
    snprintf(buf, buflen, "%s", ctx->arch);
    return IMPOP_OK;
}
