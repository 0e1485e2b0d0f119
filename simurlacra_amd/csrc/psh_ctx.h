// Context of the library and small host-side helpers shared by its translation units.
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdio>

#include "../../include/pysimhip.h"

// ------------------------------------------------------------------------------------------------ context
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
    double* tanh_tab;    // [tanh(j/256), j = 0..5120 (psh_tanh8)][tanh(j/64), j = 0..1280 (psh_tanh_tab)]
    int num_cus;
    size_t lds_per_block;
    hipEvent_t ev0, ev1;
    int timed;
    // Every rollout of a context uses the context's scratch (queue counter, weight image, plate angles, delay rings):
    // a rollout issued on another stream waits for this event, recorded behind the previous rollout's kernel.
    hipEvent_t ev_done;
    int have_done;
};

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
    *have = need;
    return PSH_OK;
}

// The scratch buffers of a rollout, grown in the context.
enum psh_scratch_kind { PSH_SCR_WFRAG = 0, PSH_SCR_PLATE = 1, PSH_SCR_DELAY = 2 };
static inline int scratch_doubles(psh_ctx* c, int which, size_t need, double** out) {
    if (which == PSH_SCR_WFRAG) { const int rc = ensure_doubles(c, &c->wfrag, &c->wfrag_doubles, need); *out = c->wfrag; return rc; }
    if (which == PSH_SCR_PLATE) { const int rc = ensure_doubles(c, &c->plate, &c->plate_doubles, need); *out = c->plate; return rc; }
    const int rc = ensure_doubles(c, &c->delay_q, &c->delay_q_doubles, need);
    *out = c->delay_q;
    return rc;
}

// Persistent launch: as many workgroups as the device keeps resident (lanes pull envs from the queue), fewer when the
// batch is smaller than that.
template <class K> static unsigned persistent_grid(psh_ctx* c, K kernel, int block, size_t lds_bytes, long n_envs) {
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, block, lds_bytes) != hipSuccess || per_cu < 1) per_cu = 1;
    const long resident = (long)per_cu * c->num_cus;
    const long need = (n_envs + block - 1) / block;
    return (unsigned)(need < resident ? need : resident);
}
