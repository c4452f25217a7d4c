// hf_rccl.hip -- the RCCL shim of the C ABI: symbols are resolved at run time from the librccl.so.1 that
// PyTorch-ROCm has already mapped (never a second copy).  Reference: the sum over data chunks of
// hessianfree/optimizer.py:677-684, across GPUs.
#include <dlfcn.h>
#include <new>

#include "hf_common.h"

struct hf_comm {
  void* comm;  // ncclComm_t
};

namespace {
struct NcclUid { char internal[128]; };
typedef int (*fn_get_uid)(NcclUid*);
typedef int (*fn_init_rank)(void**, int, NcclUid, int);
typedef int (*fn_destroy)(void*);
typedef int (*fn_allreduce)(const void*, void*, size_t, int, int, void*, hipStream_t);

void* rccl_sym(const char* name) {
  void* f = dlsym(RTLD_DEFAULT, name);
  if (f) return f;
  // torch's extension modules are loaded RTLD_LOCAL: look the library up by its
  // SONAME among the objects already mapped into this process (never load a
  // second copy).
  static void* lib = nullptr;
  if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
  if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
  if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW);
  return lib ? dlsym(lib, name) : nullptr;
}
}  // namespace

int hf_comm_unique_id(char* out128) {
  if (!out128) return HF_ERR_ARG;
  fn_get_uid f = (fn_get_uid)rccl_sym("ncclGetUniqueId");
  if (!f) return HF_ERR_NOSYMBOL;
  NcclUid id;
  const int rc = f(&id);
  if (rc) return 1000 + rc;
  memcpy(out128, id.internal, 128);
  return HF_OK;
}

int hf_comm_create(hf_comm_t** out, const char* id128, int nranks, int rank) {
  if (!out || !id128 || nranks < 1 || rank < 0 || rank >= nranks) return HF_ERR_ARG;
  fn_init_rank f = (fn_init_rank)rccl_sym("ncclCommInitRank");
  if (!f) return HF_ERR_NOSYMBOL;
  NcclUid id;
  memcpy(id.internal, id128, 128);
  hf_comm* c = new (std::nothrow) hf_comm();
  if (!c) return HF_ERR_ARG;
  const int rc = f(&c->comm, nranks, id, rank);
  if (rc) { delete c; return 1000 + rc; }
  *out = c;
  return HF_OK;
}

int hf_comm_destroy(hf_comm_t* c) {
  if (!c) return HF_OK;
  fn_destroy f = (fn_destroy)rccl_sym("ncclCommDestroy");
  if (f && c->comm) (void)f(c->comm);
  delete c;
  return HF_OK;
}

int hf_allreduce_sum(hf_comm_t* c, void* buf, int64_t n, int dtype, void* stream) {
  if (!c || !c->comm || !buf || n <= 0) return HF_ERR_ARG;
  fn_allreduce f = (fn_allreduce)rccl_sym("ncclAllReduce");
  if (!f) return HF_ERR_NOSYMBOL;
  // ncclFloat32 = 7, ncclFloat64 = 8, ncclSum = 0 (nccl.h / rccl.h enum values)
  const int nccl_dtype = dtype == HF_F32 ? 7 : 8;
  const int rc = f(buf, buf, (size_t)n, nccl_dtype, 0, c->comm, (hipStream_t)stream);
  return rc ? 1000 + rc : HF_OK;
}

int hf_allreduce_sum_multi(hf_comm_t* c, void* const* bufs, const int64_t* ns, int count, int dtype,
                           void* stream) {
  if (!c || !c->comm || !bufs || !ns || count < 1 || count > 16) return HF_ERR_ARG;
  if (dtype != HF_F32 && dtype != HF_F64) return HF_ERR_ARG;
  for (int i = 0; i < count; ++i)
    if (!bufs[i] || ns[i] <= 0) return HF_ERR_ARG;
  if (count == 1) return hf_allreduce_sum(c, bufs[0], ns[0], dtype, stream);
  typedef int (*fn_group)(void);
  fn_allreduce f = (fn_allreduce)rccl_sym("ncclAllReduce");
  fn_group gs = (fn_group)rccl_sym("ncclGroupStart");
  fn_group ge = (fn_group)rccl_sym("ncclGroupEnd");
  if (!f || !gs || !ge) return HF_ERR_NOSYMBOL;
  const int nccl_dtype = dtype == HF_F32 ? 7 : 8;
  int rc = gs();
  if (rc) return 1000 + rc;
  int first = 0;
  for (int i = 0; i < count; ++i) {
    rc = f(bufs[i], bufs[i], (size_t)ns[i], nccl_dtype, 0, c->comm, (hipStream_t)stream);
    if (rc && !first) first = rc;
  }
  rc = ge();  // (always closed: an open group would swallow every later collective of the communicator)
  if (first) return 1000 + first;
  return rc ? 1000 + rc : HF_OK;
}
