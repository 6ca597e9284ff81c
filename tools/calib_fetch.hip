// calib_fetch.hip -- calibrates rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access widths the
// engine uses (MI355X_MICROARCH.md, HBM section: "calibrate on a known byte count in your own access
// pattern").  Each kernel moves a KNOWN number of bytes over a buffer far larger than L2 + MALL:
//   rd16 : 16 B/lane coalesced loads   (k-mer pairs in kf_pass1_s, code quads in kf_pass2)
//   rd8  :  8 B/lane coalesced loads   (count quads)
//   rd1s :  1 B/lane loads, one 64-B line per wave (P / code bytes in the pass-2 drain)
//   wr1  :  1 B/lane coalesced stores  (code bytes)
//   wr8  :  8 B/lane coalesced stores  (request chunks)
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/calib_fetch tools/calib_fetch.hip
// Run  : rocprofv3 --pmc FETCH_SIZE ... -- tools/calib_fetch   (and WRITE_SIZE in a second pass)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__global__ void rd16(const ulonglong2 *p, size_t n, unsigned long long *out)
{ unsigned long long a = 0;
  for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x)
    { const ulonglong2 v = p[i]; a += v.x ^ v.y; }
  if (a == 0x1234567) *out = a;
}
__global__ void rd8(const ushort4 *p, size_t n, unsigned long long *out)
{ unsigned a = 0;
  for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x)
    { const ushort4 v = p[i]; a += v.x ^ v.y ^ v.z ^ v.w; }
  if (a == 0x1234567) *out = a;
}
__global__ void rd1s(const uint8_t *p, size_t n, unsigned long long *out)
{ unsigned a = 0;
  for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x)
    a += p[i];
  if (a == 0x1234567) *out = a;
}
__global__ void wr1(uint8_t *p, size_t n)
{ for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x)
    p[i] = (uint8_t) i;
}
__global__ void wr8(unsigned long long *p, size_t n)
{ for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x)
    p[i] = i;
}

int main()
{ const size_t bytes = 8ull << 30;                 // 8 GiB per kernel
  void *buf; unsigned long long *out;
  if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&out, 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(buf, 1, bytes);
  hipDeviceSynchronize();
  for (int rep = 0; rep < 2; rep++)
    { rd16<<<4096, 256>>>((const ulonglong2 *) buf, bytes / 16, out);
      rd8<<<4096, 256>>>((const ushort4 *) buf, bytes / 8, out);
      rd1s<<<4096, 256>>>((const uint8_t *) buf, bytes / 4, out);        // 2 GiB
      wr1<<<4096, 256>>>((uint8_t *) buf, bytes / 4);                    // 2 GiB
      wr8<<<4096, 256>>>((unsigned long long *) buf, bytes / 8);         // 8 GiB
    }
  hipDeviceSynchronize();
  printf("known bytes: rd16 %zu rd8 %zu rd1s %zu wr1 %zu wr8 %zu\n", bytes, bytes, bytes / 4, bytes / 4, bytes);
  return 0;
}
