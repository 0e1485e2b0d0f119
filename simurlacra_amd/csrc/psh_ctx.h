// Context of the library and small host-side helpers shared by its translation units.
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdio>

#include "../../include/pysimhip.h"

// ------------------------------------------------------------------------------------------------ context
#define PSH_TIMING_RING 64
struct psh_ctx {
    int device;
    char err[512];
    double* wfrag;       // packed FNN weights (fragment layout), grown on demand
    size_t wfrag_doubles;
    double* plate;       // QBB plate angles of the IK prologue [2, N], grown on demand
    size_t plate_doubles;
    double* delay_q;     // ActDelay ring buffers of the resident lanes, grown on demand
    size_t delay_q_doubles;
    unsigned int* queue; // work-queue counter of the rollout kernels
    double* tanh_tab;    // [tanh(j/256), j = -5120..5120 (the upper half is the table of psh_tanh8)][tanh(j/64), j = 0..1280 (psh_tanh_tab)]
    int num_cus;
    size_t lds_per_block;
    // HIP events around the rollout kernel of the last PSH_TIMING_RING launches (psh_rollout_ms); ev0 / ev1 = the pair of
    // the launch being issued, timed = 0 while the stream is being captured into a graph (events recorded there could
    // not be waited for)
    hipEvent_t tev0[PSH_TIMING_RING], tev1[PSH_TIMING_RING];
    long n_timed;
    hipEvent_t ev0, ev1;
    int timed;
    long n_allocs;       // device allocations made for this context's scratch so far (psh_alloc_count)
    // Every rollout of a context uses the context's scratch (queue counter, weight image, plate angles, delay rings):
    // a rollout issued on another stream waits for this event, recorded behind the previous rollout's kernel.
    hipEvent_t ev_done;
    int have_done;
    // psh_rollout_static: device address of the library's file-scope scratch array on this context's device, and
    // whether the call being issued takes its scratch from there (it then allocates nothing)
    unsigned char* static_scratch;
    int static_mode;
};

// Layout of the file-scope scratch array of psh_rollout_static (bytes): [queue counter][weight image][plate angles]
#define PSH_STATIC_QUEUE_BYTES 256
#define PSH_STATIC_WFRAG_DOUBLES ((size_t)PSH_STATIC_MAX_WEIGHT_DOUBLES)
#define PSH_STATIC_PLATE_DOUBLES (2 * (size_t)PSH_STATIC_MAX_QBB_ENVS)
#define PSH_STATIC_BYTES (PSH_STATIC_QUEUE_BYTES + 8 * (PSH_STATIC_WFRAG_DOUBLES + PSH_STATIC_PLATE_DOUBLES))

// grow-only device scratch owned by the context
static inline void set_err(psh_ctx* c, const char* fmt, const char* a = "", const char* b = "") {
    if (c) snprintf(c->err, sizeof c->err, fmt, a, b);
}

#define PSH_HIP(ctx, call)                                                    \
    do {                                                                      \
        hipError_t e_ = (call);                                               \
        if (e_ != hipSuccess) {                                               \
            set_err(ctx, "%s failed: %s", #call, hipGetErrorString(e_));      \
            return PSH_ERR_HIP;                                               \
        }                                                                     \
    } while (0)

static inline int ensure_doubles(psh_ctx* c, double** buf, size_t* have, size_t need) {
    if (need <= *have) return PSH_OK;
    if (*buf) PSH_HIP(c, hipFree(*buf));
    *buf = nullptr; *have = 0;
    PSH_HIP(c, hipMalloc(buf, need * sizeof(double)));
    c->n_allocs += 1;
    *have = need;
    return PSH_OK;
}

// The scratch buffers of a rollout.  psh_rollout grows them in the context; psh_rollout_static hands out fixed regions of
// the library's static array and rejects what does not fit.
enum psh_scratch_kind { PSH_SCR_WFRAG = 0, PSH_SCR_PLATE = 1, PSH_SCR_DELAY = 2 };
static inline int scratch_doubles(psh_ctx* c, int which, size_t need, double** out) {
    if (c->static_mode) {
        double* const base = reinterpret_cast<double*>(c->static_scratch + PSH_STATIC_QUEUE_BYTES);
        if (which == PSH_SCR_WFRAG) {
            if (need > PSH_STATIC_WFRAG_DOUBLES) { set_err(c, "psh_rollout_static: the packed policy weights exceed PSH_STATIC_MAX_WEIGHT_DOUBLES"); return PSH_ERR_UNSUPPORTED; }
            *out = base;
            return PSH_OK;
        }
        if (which == PSH_SCR_PLATE) {
            if (need > PSH_STATIC_PLATE_DOUBLES) { set_err(c, "psh_rollout_static: more than PSH_STATIC_MAX_QBB_ENVS QBallBalancer environments with the full dynamics"); return PSH_ERR_UNSUPPORTED; }
            *out = base + PSH_STATIC_WFRAG_DOUBLES;
            return PSH_OK;
        }
        set_err(c, "psh_rollout_static: the fused ActDelayWrapper needs ring buffers sized by the launch; use psh_rollout");
        return PSH_ERR_UNSUPPORTED;
    }
    if (which == PSH_SCR_WFRAG) { const int rc = ensure_doubles(c, &c->wfrag, &c->wfrag_doubles, need); *out = c->wfrag; return rc; }
    if (which == PSH_SCR_PLATE) { const int rc = ensure_doubles(c, &c->plate, &c->plate_doubles, need); *out = c->plate; return rc; }
    const int rc = ensure_doubles(c, &c->delay_q, &c->delay_q_doubles, need);
    *out = c->delay_q;
    return rc;
}

// Persistent launch: as many workgroups as the device keeps resident (lanes pull envs from the queue), fewer when the
// batch is smaller than that.
// max_per_cu > 0 limits the workgroups per CU below what fits.
template <class K> static unsigned persistent_grid(psh_ctx* c, K kernel, int block, size_t lds_bytes, long n_envs, int max_per_cu = 0) {
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, block, lds_bytes) != hipSuccess || per_cu < 1) per_cu = 1;
    if (max_per_cu > 0 && per_cu > max_per_cu) per_cu = max_per_cu;
    const long resident = (long)per_cu * c->num_cus;
    const long need = (n_envs + block - 1) / block;
    return (unsigned)(need < resident ? need : resident);
}
