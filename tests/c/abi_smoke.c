/* tests/c/abi_smoke.c -- the C ABI used from plain C (no Python, no torch, no HIP headers): what a cgo / JNI / FFI
 * binding of another host language would do.  Built and run by tests/test_gpu_parity.py::test_c_abi_from_plain_c:
 *
 *   gcc -O2 -I include -I oracle tests/c/abi_smoke.c -o abi_smoke -L discorpy_amd/lib -ldiscorpy_hip \
 *       -L oracle -lunwarp_oracle -lm  (+ rpath)
 *
 * Radial unwarp of a host image (the library stages the copies), the same through device memory obtained from
 * dcp_malloc on an explicit stream-less path, a uint16 stack chunk, and error reporting; every result is compared
 * with the CPU oracle (test infrastructure) bit for bit.  Prints "abi_smoke ok" and exits 0. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "discorpy_hip.h"
#include "unwarp_oracle.h"

#define CHECK(call)                                                              \
  do {                                                                           \
    int rc_ = (call);                                                            \
    if (rc_ != DCP_OK) {                                                         \
      fprintf(stderr, "%s -> %d: %s\n", #call, rc_, dcp_last_error());           \
      return 1;                                                                  \
    }                                                                            \
  } while (0)

static uint32_t lcg(uint32_t *s) { *s = *s * 1664525u + 1013904223u; return *s >> 8; }

int main(void) {
  const int64_t H = 300, W = 421;
  const double fact[5] = {1.002, -3e-5, 9e-8, -1.5e-10, 8e-14};
  const double xc = 201.3, yc = 140.8;
  uint32_t seed = 12345u;
  float *img = malloc(sizeof(float) * H * W), *got = malloc(sizeof(float) * H * W), *want = malloc(sizeof(float) * H * W);
  for (int64_t i = 0; i < H * W; ++i) img[i] = (float)lcg(&seed) / 16777216.0f;
  if (dcp_device_count() < 1) { fprintf(stderr, "no HIP device\n"); return 2; }

  /* 1. host pointers */
  CHECK(dcp_unwarp_image_f32(img, got, H, W, W, 1, xc, yc, fact, 5, 1, 1, DCP_BLEND_SCIPY, DCP_MEM_HOST, -1, NULL));
  if (orc_unwarp_image_f32(img, want, H, W, W, xc, yc, fact, 5, 1, 1, ORC_POLY_KERNEL, ORC_BLEND_SCIPY) != 0) return 3;
  if (memcmp(got, want, sizeof(float) * H * W) != 0) { fprintf(stderr, "host path differs from the oracle\n"); return 4; }

  /* 2. device pointers from the library's own allocator, default blend */
  void *dsrc = NULL, *ddst = NULL;
  CHECK(dcp_malloc(&dsrc, sizeof(float) * H * W, 0));
  CHECK(dcp_malloc(&ddst, sizeof(float) * H * W, 0));
  CHECK(dcp_memcpy(dsrc, img, sizeof(float) * H * W, DCP_COPY_H2D, 0, NULL));
  CHECK(dcp_unwarp_image_f32(dsrc, ddst, H, W, W, 1, xc, yc, fact, 5, 1, 1, DCP_BLEND_F64LERP, DCP_MEM_DEVICE, 0, NULL));
  CHECK(dcp_memcpy(got, ddst, sizeof(float) * H * W, DCP_COPY_D2H, 0, NULL));
  if (orc_unwarp_image_f32(img, want, H, W, W, xc, yc, fact, 5, 1, 1, ORC_POLY_KERNEL, ORC_BLEND_F64LERP) != 0) return 5;
  if (memcmp(got, want, sizeof(float) * H * W) != 0) { fprintf(stderr, "device path differs from the oracle\n"); return 6; }
  CHECK(dcp_free(dsrc, 0));
  CHECK(dcp_free(ddst, 0));

  /* 3. rows 40..59 of a uint16 stack (unwarp_chunk_slices_backward semantics), host memory */
  const int64_t D = 3, R0 = 40, NR = 20;
  uint16_t *vol = malloc(sizeof(uint16_t) * D * H * W), *sino = malloc(sizeof(uint16_t) * D * NR * W);
  uint16_t *ref = malloc(sizeof(uint16_t) * NR * W);
  for (int64_t i = 0; i < D * H * W; ++i) vol[i] = (uint16_t)(lcg(&seed) & 0xffff);
  CHECK(dcp_unwarp_stack_rows_typed(vol, sino, DCP_DTYPE_U16, 0, D, H, W, H * W, W, xc, yc, fact, 5, (double)R0, NR, 1,
                                    DCP_MEM_HOST, -1, NULL));
  double *yd = malloc(sizeof(double) * H * W), *xd = malloc(sizeof(double) * H * W);
  if (orc_radial_coords(H, W, xc, yc, fact, 5, ORC_POLY_KERNEL, 1, yd, xd) != 0) return 7;
  for (int64_t d = 0; d < D; ++d) {
    if (orc_map_coordinates_typed(vol + d * H * W, ref, ORC_DT_U16, H, W, W, yd + R0 * W, xd + R0 * W, 1, NR * W, 1,
                                  ORC_MODE_REFLECT, NULL) != 0) return 8;
    if (memcmp(sino + d * NR * W, ref, sizeof(uint16_t) * NR * W) != 0) { fprintf(stderr, "uint16 stack differs (projection %lld)\n", (long long)d); return 9; }
  }

  /* 4. errors come back as codes + a message */
  if (dcp_unwarp_image_f32(img, got, H, W, W, 1, xc, yc, fact, 99, 1, 1, DCP_BLEND_SCIPY, DCP_MEM_HOST, -1, NULL) != DCP_ERR_INVALID_ARG ||
      strstr(dcp_last_error(), "nfact") == NULL) { fprintf(stderr, "expected an nfact error, got: %s\n", dcp_last_error()); return 10; }
  printf("abi_smoke ok (library version %d, %d device(s))\n", dcp_version(), dcp_device_count());
  return 0;
}
