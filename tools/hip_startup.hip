// hip_startup.hip -- what the HIP runtime costs a process before its first kernel runs (the floor under `hetmers`' start-up):
//   hipcc --offload-arch=gfx950 -O2 -o tools/hip_startup tools/hip_startup.hip ; tools/hip_startup [path/to/libsmg_hetmers.so]
// prints: runtime init (first API call), first launch of a kernel of THIS (tiny) module, and -- with a library path -- dlopen of
// the engine plus smg_engine_create + a first launch through it (its 5.5 MB code object is loaded then).
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <stdio.h>
#include <time.h>
static double now() { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e3 + t.tv_nsec * 1e-6; }
__global__ void k_nop(int *p) { if (p) *p = 1; }
int main(int argc, char **argv)
{ double t0 = now();
  hipSetDevice(0); hipFree(0);
  double t1 = now();
  hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, 0, (int *) 0); hipDeviceSynchronize();
  double t2 = now();
  void *buf = 0; hipMalloc(&buf, 1 << 20); hipMemset(buf, 0, 1 << 20); hipDeviceSynchronize();
  double t3 = now();
  printf("runtime init %.1f ms, first launch (tiny module) %.1f ms, first malloc+memset %.1f ms\n", t1 - t0, t2 - t1, t3 - t2);
  if (argc > 1)
    { double a = now();
      void *h = dlopen(argv[1], RTLD_NOW);
      double b = now();
      if (!h) { printf("dlopen failed: %s\n", dlerror()); return 1; }
      typedef void *(*create_t)(int, void *, char *, size_t);
      typedef int (*bind_t)(void *, int, long long, const void *, const void *, char *, size_t);
      typedef int (*run_t)(void *, int, void *, void *, char *, size_t);
      create_t create = (create_t) dlsym(h, "smg_engine_create");
      bind_t bind = (bind_t) dlsym(h, "smg_engine_bind");
      run_t run = (run_t) dlsym(h, "smg_engine_run");
      char err[256];
      void *e = create(0, NULL, err, sizeof(err));
      double c = now();
      unsigned long long *keys = 0; unsigned short *cnt = 0; long long *plot = 0;
      hipMalloc(&keys, 4096 * 8); hipMalloc(&cnt, 4096 * 2 + 64); hipMalloc(&plot, 8 * 1001 * 501);
      unsigned long long hk[4096]; unsigned short hc[4096];
      for (int i = 0; i < 4096; i++) { hk[i] = ((unsigned long long) i * 0x9E3779B97F4A7C15ull) | 1ull; hc[i] = 20; }
      // (sorted keys: multiples of an odd constant are not sorted -- sort by insertion of the high bits)
      for (int i = 0; i < 4096; i++) hk[i] = ((unsigned long long) i << 50) << 2;
      hipMemcpy(keys, hk, sizeof(hk), hipMemcpyHostToDevice); hipMemcpy(cnt, hc, sizeof(hc), hipMemcpyHostToDevice);
      int rc = bind(e, 31, 4096, keys, cnt, err, sizeof(err));
      double d = now();
      rc = rc ? rc : run(e, 2 /* SMG_SYM_NONE: general path */, plot, NULL, err, sizeof(err));
      double f = now();
      printf("engine: dlopen %.1f ms, create %.1f ms, bind %.1f ms, first run (code object load + kernels) %.1f ms rc=%d\n", b - a, c - b, d - c, f - d, rc);
    }
  return 0;
}
