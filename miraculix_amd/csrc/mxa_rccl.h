// mxa_rccl.h -- RCCL as mxa_multi.cpp uses it: bound at run time (dlopen; the library is not linked and the default reduction does not need it), TYPED at
// compile time.  The prototypes and enumerators come from the image's own <rccl/rccl.h>: every pointer below has the type of the header's declaration, so a
// header whose signatures differ from what mxa_multi.cpp calls does not compile, and the static_asserts pin the values that DESIGN.md section 6 records
// (ncclFloat64 = 8, ncclSum = 0, ncclSuccess = 0).  tests/test_abi_cpu.py compiles this header with a wrong expectation (-DMXA_RCCL_EXPECT_FLOAT64=7) and
// requires the compiler to refuse it.
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <type_traits>

#ifndef MXA_RCCL_EXPECT_FLOAT64
#define MXA_RCCL_EXPECT_FLOAT64 8
#endif
#ifndef MXA_RCCL_EXPECT_SUM
#define MXA_RCCL_EXPECT_SUM 0
#endif

namespace mxa {

struct Rccl {
  void *lib = nullptr;
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclReduce) Reduce = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  bool ok = false;
};
static_assert(std::is_same<decltype(&ncclReduce), ncclResult_t (*)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t)>::value,
              "rccl.h: ncclReduce is not (sendbuff, recvbuff, count, datatype, op, root, comm, stream)");
static_assert(std::is_same<decltype(&ncclCommInitAll), ncclResult_t (*)(ncclComm_t *, int, const int *)>::value, "rccl.h: ncclCommInitAll is not (comm[], ndev, devlist)");
static_assert(std::is_same<decltype(&ncclCommDestroy), ncclResult_t (*)(ncclComm_t)>::value, "rccl.h: ncclCommDestroy is not (comm)");
static_assert(std::is_same<decltype(&ncclGroupStart), ncclResult_t (*)()>::value && std::is_same<decltype(&ncclGroupEnd), ncclResult_t (*)()>::value, "rccl.h: ncclGroupStart / ncclGroupEnd take arguments");
static_assert((int)ncclFloat64 == MXA_RCCL_EXPECT_FLOAT64 && (int)ncclSum == MXA_RCCL_EXPECT_SUM && (int)ncclSuccess == 0, "rccl.h: enumerator values differ from the recorded ones");

inline Rccl &rccl() {
  static Rccl r = [] {
    Rccl q;
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      q.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (q.lib) break;
    }
    if (!q.lib) return q;
    q.CommInitAll = reinterpret_cast<decltype(q.CommInitAll)>(dlsym(q.lib, "ncclCommInitAll"));
    q.CommDestroy = reinterpret_cast<decltype(q.CommDestroy)>(dlsym(q.lib, "ncclCommDestroy"));
    q.Reduce = reinterpret_cast<decltype(q.Reduce)>(dlsym(q.lib, "ncclReduce"));
    q.GroupStart = reinterpret_cast<decltype(q.GroupStart)>(dlsym(q.lib, "ncclGroupStart"));
    q.GroupEnd = reinterpret_cast<decltype(q.GroupEnd)>(dlsym(q.lib, "ncclGroupEnd"));
    q.GetErrorString = reinterpret_cast<decltype(q.GetErrorString)>(dlsym(q.lib, "ncclGetErrorString"));
    q.ok = q.CommInitAll && q.CommDestroy && q.Reduce && q.GroupStart && q.GroupEnd;
    return q;
  }();
  return r;
}

}  // namespace mxa
