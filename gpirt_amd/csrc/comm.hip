// libgpirt_rccl.so -- the collective of the item-sharded respondent step on RCCL, without torch.
//
// The respondent step needs, once per lock-step proposal round, the sum over ranks of every
// rank's per-respondent partial log-likelihoods (an n-vector, or 4 n with four proposals per
// round).  Here that sum is an all-gather of the ranks' vectors over xGMI followed by a local
// sum in rank order: every rank adds the same numbers in the same order, so all ranks hold the
// same bits and take the same accept / shrink decisions, run after run.
// Kept apart from libgpirt_hip.so so that single-GPU users never load RCCL.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "../../include/gpirt_comm.h"

static thread_local char g_comm_err[512] = "";

static void comm_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_comm_err, sizeof g_comm_err, fmt, ap);
  va_end(ap);
}

#define CM_HIP(expr)                                                                          \
  do {                                                                                        \
    hipError_t e_ = (expr);                                                                   \
    if (e_ != hipSuccess) {                                                                   \
      comm_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__);  \
      return GPIRT_ERR_HIP;                                                                   \
    }                                                                                         \
  } while (0)
#define CM_NCCL(expr)                                                                         \
  do {                                                                                        \
    ncclResult_t r_ = (expr);                                                                 \
    if (r_ != ncclSuccess) {                                                                  \
      comm_error("%s failed: %s (%s:%d)", #expr, ncclGetErrorString(r_), __FILE__, __LINE__); \
      return GPIRT_ERR_COMM;                                                                  \
    }                                                                                         \
  } while (0)

struct gpirt_comm {
  ncclComm_t nccl;
  int rank, world, device;
  hipStream_t stream;
  int have_stream;
  double *gather;        // [world, capacity] device staging of the all-gather
  int64_t capacity;      // doubles per rank the staging holds
  int64_t calls;
};

// buf[i] = g[i] + g[count + i] + ... + g[(world-1) count + i], ranks in order
__global__ void sum_ranks_kernel(const double *__restrict__ g, int world, int64_t count,
                                 double *__restrict__ buf) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  double acc = g[i];
  for (int r = 1; r < world; ++r) acc += g[(int64_t)r * count + i];
  buf[i] = acc;
}

extern "C" {

const char *gpirt_comm_last_error(void) { return g_comm_err; }

int gpirt_comm_unique_id(void *id) {
  if (!id) { comm_error("gpirt_comm_unique_id: id is NULL"); return GPIRT_ERR_BAD_ARG; }
  static_assert(GPIRT_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
  ncclUniqueId u;
  CM_NCCL(ncclGetUniqueId(&u));
  memcpy(id, &u, sizeof u);
  return GPIRT_OK;
}

int gpirt_comm_create(const void *id, int32_t rank, int32_t world, gpirt_comm **out) {
  if (!id || !out || world < 1 || rank < 0 || rank >= world) {
    comm_error("gpirt_comm_create: bad argument (rank %d of %d)", rank, world);
    return GPIRT_ERR_BAD_ARG;
  }
  gpirt_comm *c = new gpirt_comm();
  c->rank = rank; c->world = world;
  if (hipGetDevice(&c->device) != hipSuccess) { delete c; comm_error("no current HIP device"); return GPIRT_ERR_NO_DEVICE; }
  ncclUniqueId u;
  memcpy(&u, id, sizeof u);
  ncclResult_t r = ncclCommInitRank(&c->nccl, world, u, rank);
  if (r != ncclSuccess) {
    comm_error("ncclCommInitRank(rank %d of %d) failed: %s", rank, world, ncclGetErrorString(r));
    delete c;
    return GPIRT_ERR_COMM;
  }
  *out = c;
  return GPIRT_OK;
}

int gpirt_comm_set_stream(gpirt_comm *c, void *stream) {
  if (!c) { comm_error("gpirt_comm_set_stream: comm is NULL"); return GPIRT_ERR_BAD_ARG; }
  c->stream = (hipStream_t)stream;
  c->have_stream = 1;
  return GPIRT_OK;
}

// room for `count` doubles per rank in the all-gather's staging
static int comm_reserve(gpirt_comm *c, int64_t count) {
  if (count <= c->capacity) return GPIRT_OK;
  CM_HIP(hipSetDevice(c->device));
  // what is in flight on the stream may still read the old staging
  if (c->have_stream) CM_HIP(hipStreamSynchronize(c->stream));
  if (c->gather) CM_HIP(hipFree(c->gather));
  c->gather = nullptr;
  c->capacity = 0;
  CM_HIP(hipMalloc((void **)&c->gather, sizeof(double) * (size_t)count * c->world));
  c->capacity = count;
  return GPIRT_OK;
}

int gpirt_comm_allreduce(double *d_buf, int64_t count, void *user) {
  gpirt_comm *c = (gpirt_comm *)user;
  if (!c || !d_buf || count < 1) { comm_error("gpirt_comm_allreduce: bad argument"); return GPIRT_ERR_BAD_ARG; }
  if (!c->have_stream) {
    comm_error("gpirt_comm_allreduce: no stream set (call gpirt_comm_set_stream with gpirt_sampler_stream() "
               "after creating the sampler)");
    return GPIRT_ERR_BAD_ARG;
  }
  CM_HIP(hipSetDevice(c->device));
  // gpirt_comm_bind sized the staging for the sampler's largest exchange; a caller that binds the hook by
  // hand (or changes the configuration after binding) still gets a staging that grows, at the price of a
  // stream drain and an allocation inside the sweep
  if (count > c->capacity) {
    const int rc = comm_reserve(c, count);
    if (rc != GPIRT_OK) return rc;
  }
  CM_NCCL(ncclAllGather(d_buf, c->gather, (size_t)count, ncclDouble, c->nccl, c->stream));
  hipLaunchKernelGGL(sum_ranks_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, c->stream,
                     c->gather, c->world, count, d_buf);
  CM_HIP(hipGetLastError());
  c->calls++;
  return GPIRT_OK;
}

int gpirt_comm_bind(gpirt_comm *c, gpirt_config *cfg) {
  if (!c || !cfg) { comm_error("gpirt_comm_bind: NULL argument"); return GPIRT_ERR_BAD_ARG; }
  cfg->rank = c->rank;
  cfg->world = c->world;
  cfg->allreduce = gpirt_comm_allreduce;
  cfg->allreduce_user = c;
  cfg->flags |= GPIRT_ALLREDUCE_ON_STREAM;
  // the largest vector a sweep of this configuration exchanges, so that no call inside a sweep allocates:
  // four proposals per respondent and round; with GPIRT_FIXED_ORDER_SUM one such vector per block of items
  // (the blocks of the bucketed evaluation, at most GPIRT_SUM_BLOCK items); the static respondent step
  // exchanges the padded grid x respondents table
  int64_t count = 4 * (int64_t)cfg->n;
  if (cfg->flags & GPIRT_FIXED_ORDER_SUM) {
    const int C = cfg->C > 0 ? cfg->C : 2;
    int chunk = (6144 / C) & ~63;
    chunk = chunk < 64 ? 64 : (chunk > GPIRT_SUM_BLOCK ? GPIRT_SUM_BLOCK : chunk);
    count *= (cfg->m + chunk - 1) / chunk;
  }
  if (cfg->theta_mode == GPIRT_THETA_GRID) count = (int64_t)((cfg->n_grid + 127) / 128 * 128) * cfg->n;
  return comm_reserve(c, count);
}

int64_t gpirt_comm_calls(const gpirt_comm *c) { return c ? c->calls : 0; }

int gpirt_comm_world(const gpirt_comm *c) {
  if (!c) return 0;
  int count = 0;
  if (ncclCommCount(c->nccl, &count) != ncclSuccess) return 0;
  return count;
}

int64_t gpirt_comm_capacity(const gpirt_comm *c) { return c ? c->capacity : 0; }

void gpirt_comm_destroy(gpirt_comm *c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->have_stream) (void)hipStreamSynchronize(c->stream);
  if (c->gather) (void)hipFree(c->gather);
  (void)ncclCommDestroy(c->nccl);
  delete c;
}

}  // extern "C"
