// vmm_probe.cpp -- cold-start probe (round 5): what a fresh process pays the driver for a LARGE device buffer through
//   (a) one hipMalloc, (b) a virtual address reservation with 1 GB physical chunks mapped into it (hipMemCreate / hipMemMap),
// first and second touch included. Build: hipcc --offload-arch=gfx950 -O2 vmm_probe.cpp -o vmm_probe.bin
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("{\"error\": \"%s at %s\"}\n", hipGetErrorString(e_), #x); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
  const size_t GB = (size_t)1 << 30;
  const size_t S = argc > 1 ? (size_t)atoll(argv[1]) : 40;
  const char* mode = argc > 2 ? argv[2] : "vmm";
  const size_t chunk_gb = argc > 3 ? (size_t)atoll(argv[3]) : 1;
  CK(hipFree(nullptr));
  double t0 = now();
  void* p = nullptr;
  std::vector<hipMemGenericAllocationHandle_t> hs;
  if (mode[0] == 'v') {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    CK(hipMemAddressReserve(&p, S * GB, 0, nullptr, 0));
    const size_t chunk = chunk_gb * GB;
    for (size_t off = 0; off < S * GB; off += chunk) {
      hipMemGenericAllocationHandle_t h;
      CK(hipMemCreate(&h, chunk, &prop, 0));
      CK(hipMemMap((char*)p + off, chunk, 0, h, 0));
      hs.push_back(h);
    }
    hipMemAccessDesc acc = {};
    acc.location.type = hipMemLocationTypeDevice;
    acc.location.id = 0;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(p, S * GB, &acc, 1));
    printf("{\"granularity\": %zu, ", gran);
  } else {
    CK(hipMalloc(&p, S * GB));
    printf("{");
  }
  CK(hipDeviceSynchronize());
  double t1 = now();
  CK(hipMemset(p, 0, S * GB));
  CK(hipDeviceSynchronize());
  double t2 = now();
  CK(hipMemset(p, 1, S * GB));
  CK(hipDeviceSynchronize());
  double t3 = now();
  printf("\"GB\": %zu, \"mode\": \"%s\", \"chunk_gb\": %zu, \"alloc_s\": %.4f, \"first_touch_s\": %.4f, \"second_touch_s\": %.4f}\n", S, mode,
         chunk_gb, t1 - t0, t2 - t1, t3 - t2);
  return 0;
}
