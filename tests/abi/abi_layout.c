/* abi_layout.c -- prints sizeof / offsetof of every struct include/curvis_hip.h declares, one line per field:
 *   <struct> <field> <offset> <size>      (field "." = the struct itself, offset = its alignment)
 * tests/test_abi.py compiles it with the host C compiler and compares the numbers with the table next to the
 * #[repr(C)] structs in INTEGRATION.md and with the ctypes structures in curvis_amd/_abi.py, so that the Rust
 * stub (which cannot be compiled in this image: no rustc) is checked field by field against the C header. */
#include <stddef.h>
#include <stdio.h>

#include "../../include/curvis_hip.h"

#define S(T) printf(#T " . %zu %zu\n", _Alignof(T), sizeof(T))
#define F(T, f) printf(#T " " #f " %zu %zu\n", offsetof(T, f), sizeof(((T *)0)->f))

int main(void) {
  S(curvis_metric);
  F(curvis_metric, kind);
  F(curvis_metric, _pad);
  F(curvis_metric, rho);
  F(curvis_metric, m);
  F(curvis_metric, a);
  S(curvis_camera);
  F(curvis_camera, pos);
  F(curvis_camera, rot);
  F(curvis_camera, focal);
  F(curvis_camera, sensor_w);
  F(curvis_camera, sensor_h);
  F(curvis_camera, res_x);
  F(curvis_camera, res_y);
  S(curvis_ray_debug);
  F(curvis_ray_debug, x);
  F(curvis_ray_debug, p);
  F(curvis_ray_debug, steps);
  F(curvis_ray_debug, code);
  F(curvis_ray_debug, tx);
  F(curvis_ray_debug, ty);
  S(curvis_stats);
  F(curvis_stats, rays);
  F(curvis_stats, steps);
  F(curvis_stats, n_pos);
  F(curvis_stats, n_neg);
  F(curvis_stats, n_none);
  F(curvis_stats, n_oob);
  F(curvis_stats, kernel_ms);
  F(curvis_stats, total_ms);
  F(curvis_stats, integrate_ms);
  F(curvis_stats, shade_ms);
  S(curvis_sampling_info);
  F(curvis_sampling_info, n_samples);
  F(curvis_sampling_info, rounds);
  F(curvis_sampling_info, calls);
  F(curvis_sampling_info, steps);
  F(curvis_sampling_info, warned_max_iterations);
  F(curvis_sampling_info, _pad);
  printf("CURVIS_ABI_VERSION . %d 0\n", CURVIS_ABI_VERSION);
  return 0;
}
