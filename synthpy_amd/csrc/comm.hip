// Ray-sharded multi-GPU: the one collective of the path, a sum of the per-GPU detector images
// (reference: comm.reduce(sh.H, root=0, op=MPI.SUM), examples/jobs/run_scripts/pvti_trace_mpi.py:169-170).
// RCCL over xGMI, one rank per GPU.  librccl is bound at first use (dlopen), so single-GPU use of
// libsynthray.so does not load it.  uint32 counts reduce exactly (order-independent); the complex
// image reduces in float64.
#include <dlfcn.h>

#include <cstring>
#include <rccl/rccl.h>

#include <algorithm>

#include "common.hpp"

struct sr_comm {
  ncclComm_t comm = nullptr;
  int rank = 0, n_ranks = 1;
};

namespace {

struct Rccl {
  void *h = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*Reduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
  ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

int rccl(Rccl **out) {
  static Rccl R;
  if (!R.h) {
    const char *names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char *n : names) {
      R.h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
      if (R.h) break;
    }
    if (!R.h) return sr::fail(SR_ERR_RCCL, "cannot load librccl.so: %s", dlerror());
#define SR_SYM(field, name)                                                       \
  R.field = reinterpret_cast<decltype(R.field)>(dlsym(R.h, name));                \
  if (!R.field) return sr::fail(SR_ERR_RCCL, "librccl.so lacks symbol %s", name);
    SR_SYM(GetUniqueId, "ncclGetUniqueId")
    SR_SYM(CommInitRank, "ncclCommInitRank")
    SR_SYM(CommDestroy, "ncclCommDestroy")
    SR_SYM(Reduce, "ncclReduce")
    SR_SYM(AllReduce, "ncclAllReduce")
    SR_SYM(Send, "ncclSend")
    SR_SYM(Recv, "ncclRecv")
    SR_SYM(CommCount, "ncclCommCount")
    SR_SYM(CommUserRank, "ncclCommUserRank")
    SR_SYM(GetErrorString, "ncclGetErrorString")
#undef SR_SYM
  }
  *out = &R;
  return SR_OK;
}

static_assert(sizeof(ncclUniqueId) == SR_COMM_ID_BYTES, "ncclUniqueId is expected to be 128 bytes");

}  // namespace

extern "C" {

int sr_comm_unique_id(void *id128) {
  SR_CHECK(id128 != nullptr, "sr_comm_unique_id: NULL buffer");
  Rccl *R;
  int rc = rccl(&R);
  if (rc) return rc;
  ncclUniqueId id;
  ncclResult_t e = R->GetUniqueId(&id);
  if (e != ncclSuccess) return sr::fail(SR_ERR_RCCL, "ncclGetUniqueId: %s", R->GetErrorString(e));
  std::memcpy(id128, &id, sizeof(id));
  return SR_OK;
}

int sr_comm_create(sr_comm **out, const void *id128, int rank, int n_ranks) {
  SR_CHECK(out && id128, "sr_comm_create: NULL argument");
  *out = nullptr;
  SR_CHECK(n_ranks >= 1 && rank >= 0 && rank < n_ranks, "sr_comm_create: rank %d of %d", rank, n_ranks);
  int rc = sr::ensure_init();
  if (rc) return rc;
  Rccl *R;
  if ((rc = rccl(&R))) return rc;
  ncclUniqueId id;
  std::memcpy(&id, id128, sizeof(id));
  sr_comm *c = new sr_comm();
  c->rank = rank;
  c->n_ranks = n_ranks;
  ncclResult_t e = R->CommInitRank(&c->comm, n_ranks, id, rank);
  if (e != ncclSuccess) {
    delete c;
    return sr::fail(SR_ERR_RCCL, "ncclCommInitRank: %s", R->GetErrorString(e));
  }
  *out = c;
  return SR_OK;
}

int sr_image_reduce(sr_image *img, sr_comm *comm, int root) {
  SR_CHECK(img && comm, "sr_image_reduce: NULL argument");
  SR_CHECK(root < comm->n_ranks, "sr_image_reduce: root %d out of range", root);
  Rccl *R;
  int rc = rccl(&R);
  if (rc) return rc;
  hipStream_t st = sr::ctx().stream;
  const bool counts = img->kind == SR_IMG_COUNTS;
  const size_t n = counts ? (size_t)img->bytes / sizeof(uint32_t) : (size_t)img->bytes / sizeof(double);
  const ncclDataType_t dt = counts ? ncclUint32 : ncclFloat64;
  ncclResult_t e = root < 0 ? R->AllReduce(img->d, img->d, n, dt, ncclSum, comm->comm, st)
                            : R->Reduce(img->d, img->d, n, dt, ncclSum, root, comm->comm, st);
  if (e != ncclSuccess) return sr::fail(SR_ERR_RCCL, "RCCL reduce: %s", R->GetErrorString(e));
  return SR_OK;
}

int sr_comm_ranks(const sr_comm *comm, int *rank, int *n_ranks) {
  SR_CHECK(comm && rank && n_ranks, "sr_comm_ranks: NULL argument");
  Rccl *R;
  int rc = rccl(&R);
  if (rc) return rc;
  ncclResult_t e = R->CommUserRank(comm->comm, rank);
  if (e == ncclSuccess) e = R->CommCount(comm->comm, n_ranks);
  if (e != ncclSuccess) return sr::fail(SR_ERR_RCCL, "ncclCommUserRank / ncclCommCount: %s", R->GetErrorString(e));
  return SR_OK;
}

// A12: the ray records of a slab-decomposed trace go to the GPU that holds the next slab, point to point over xGMI
int sr_rays_handoff_send(sr_rays *r, sr_comm *comm, int peer) {
  SR_CHECK(r && comm, "sr_rays_handoff_send: NULL argument");
  SR_CHECK(peer >= 0 && peer < comm->n_ranks && peer != comm->rank, "sr_rays_handoff_send: peer %d", peer);
  if (!r->have_rec) return sr::fail(SR_ERR_STATE, "sr_rays_handoff_send: no hand-off records (trace with SR_HANDOFF_EXIT first)");
  if (r->n == 0) return SR_OK;
  Rccl *R;
  int rc = rccl(&R);
  if (rc) return rc;
  ncclResult_t e = R->Send(r->rec, (size_t)10 * r->n, ncclFloat64, peer, comm->comm, sr::ctx().stream);
  if (e != ncclSuccess) return sr::fail(SR_ERR_RCCL, "ncclSend: %s", R->GetErrorString(e));
  return SR_OK;
}

int sr_rays_handoff_recv(sr_rays *r, sr_comm *comm, int peer) {
  SR_CHECK(r && comm, "sr_rays_handoff_recv: NULL argument");
  SR_CHECK(peer >= 0 && peer < comm->n_ranks && peer != comm->rank, "sr_rays_handoff_recv: peer %d", peer);
  if (r->n == 0) return SR_OK;
  if (!r->rec) {  // by the bundle's capacity, as every buffer allocated after sr_rays_create (common.hpp)
    int rc = sr::dev_alloc(&r->rec, (size_t)10 * (size_t)std::max(r->cap, r->n));
    if (rc) return rc;
  }
  Rccl *R;
  int rc = rccl(&R);
  if (rc) return rc;
  ncclResult_t e = R->Recv(r->rec, (size_t)10 * r->n, ncclFloat64, peer, comm->comm, sr::ctx().stream);
  if (e != ncclSuccess) return sr::fail(SR_ERR_RCCL, "ncclRecv: %s", R->GetErrorString(e));
  r->have_rec = true;
  r->traced = false;
  if (!r->bbox_given) r->have_bbox = false;  // other rays than the bundle's last upload: judged by the whole lateral grid, unless the caller named their beam
  return SR_OK;
}

void sr_comm_destroy(sr_comm *comm) {
  if (!comm) return;
  Rccl *R;
  if (comm->comm && rccl(&R) == SR_OK) R->CommDestroy(comm->comm);
  delete comm;
}

}  // extern "C"
