#include "msm_unit.h"
CELO_DEFINE_MSM_UNIT(celo::G1_377, g1_377)

// group-agnostic ends of a fixed-base handle (msm.h FixedTable; capi.hip celo_amd_msm_fixed_release / _info)
namespace celo {
int fixed_table_release(FixedTable* T) {
  if (!T) return 2;
  if (int rc = api_enter()) return rc;
  int prev = 0;
  (void)hipGetDevice(&prev);
  (void)hipSetDevice(T->device);
  if (T->table) (void)hipFree(T->table);
  if (T->tinf) (void)hipFree(T->tinf);
  (void)hipSetDevice(prev);
  delete T;
  return 0;
}
int fixed_table_info(const FixedTable* T, size_t* n, int* window_bits, int* windows, size_t* table_bytes, float* build_ms) {
  if (!T) return 2;
  if (n) *n = T->n;
  if (window_bits) *window_bits = T->cf;
  if (windows) *windows = (int)T->W;
  if (table_bytes) *table_bytes = T->bytes;
  if (build_ms) *build_ms = T->build_ms;
  return 0;
}
}  // namespace celo
