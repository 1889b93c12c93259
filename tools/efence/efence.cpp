// Debugging allocator for PyTorch-ROCm ("electric fence"): every device tensor gets its own virtual-memory mapping and sits at
// the END of it, with an unmapped granule behind -- a kernel that reads or writes past the end of a buffer faults at once instead
// of touching a neighbour.  Used with tools/efence/run_pytest.py (serialised launches: the Python frame of the faulting op shows).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>

namespace {
struct Rec { void* va; size_t reserved, mapped; hipMemGenericAllocationHandle_t h; };
std::map<void*, Rec> recs;
std::mutex mu;
size_t rup(size_t x, size_t a) { return (x + a - 1) / a * a; }
#define EF_CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "[efence] %s -> %s\n", #x, hipGetErrorString(e_)); abort(); } } while (0)
}

extern "C" void* ef_malloc(ssize_t size, int device, hipStream_t) {
  if (size <= 0) size = 16;
  hipMemAllocationProp prop{};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = device;
  size_t gran = 0;
  EF_CHECK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum));
  const size_t mapped = rup((size_t)size, gran), reserved = mapped + gran;
  void* va = nullptr;
  EF_CHECK(hipMemAddressReserve(&va, reserved, gran, nullptr, 0));
  hipMemGenericAllocationHandle_t h;
  EF_CHECK(hipMemCreate(&h, mapped, &prop, 0));
  EF_CHECK(hipMemMap(va, mapped, 0, h, 0));
  hipMemAccessDesc acc{};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  EF_CHECK(hipMemSetAccess(va, mapped, &acc, 1));
  // EF_ALIGN (default 256, what torch's own allocator guarantees at least): alignment of the returned pointer; 16 also catches
  // overreads shorter than 256 bytes but hands out bases no real allocation has
  static const size_t align = getenv("EF_ALIGN") ? (size_t)atoi(getenv("EF_ALIGN")) : 256;
  void* p = static_cast<char*>(va) + mapped - rup((size_t)size, align);
  std::lock_guard<std::mutex> g(mu);
  recs[p] = Rec{va, reserved, mapped, h};
  return p;
}

extern "C" void ef_free(void* p, ssize_t, int, hipStream_t) {
  // EF_UNMAP=1 also unmaps freed tensors (use-after-free faults too); off by default: on this ROCm, tests that allocate and free
  // in quick succession then read stale translations -- wrong values and faults that no other allocator reproduces -- so a freed
  // tensor simply stays mapped (short debugging runs only: nothing is ever given back)
  static const bool unmap = getenv("EF_UNMAP") && atoi(getenv("EF_UNMAP")) == 1;
  if (!p || !unmap) return;
  Rec r;
  {
    std::lock_guard<std::mutex> g(mu);
    auto it = recs.find(p);
    if (it == recs.end()) return;
    r = it->second;
    recs.erase(it);
  }
  (void)hipDeviceSynchronize();
  EF_CHECK(hipMemUnmap(r.va, r.mapped));
  EF_CHECK(hipMemRelease(r.h));
  EF_CHECK(hipMemAddressFree(r.va, r.reserved));
}
