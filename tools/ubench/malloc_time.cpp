// hipMalloc / hipFree wall clock by size (round 6: what the first report call of a matrix pays for its N-sized scratch arrays)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
int main() {
  void* w; hipMalloc(&w, 1 << 20); hipMemset(w, 0, 1 << 20); hipDeviceSynchronize();
  for (int rep = 0; rep < 2; ++rep)
    for (size_t mb : {1, 12, 35, 100, 200, 400, 800}) {
      auto t0 = std::chrono::steady_clock::now();
      void* p = nullptr; hipMalloc(&p, mb << 20);
      auto t1 = std::chrono::steady_clock::now();
      hipMemsetAsync(p, 0, mb << 20, 0); hipDeviceSynchronize();
      auto t2 = std::chrono::steady_clock::now();
      hipFree(p);
      auto t3 = std::chrono::steady_clock::now();
      printf("%4zu MB  hipMalloc %7.3f ms  first touch (memset) %7.3f ms  hipFree %7.3f ms\n", mb, std::chrono::duration<double, std::milli>(t1 - t0).count(),
             std::chrono::duration<double, std::milli>(t2 - t1).count(), std::chrono::duration<double, std::milli>(t3 - t2).count());
    }
  return 0;
}
