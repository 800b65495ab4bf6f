// Temporary: entry points whose kernels are not written yet.
#include "csx_internal.h"
namespace csx {
void free_triplan(TriPlan *) {}
void free_cholplan(CholPlan *) {}
}
extern "C" {
int csx_multiply(csx_handle_t, csx_handle_t, csx_handle_t *) { return CSX_EINVAL; }
int csx_tri_analyse(csx_handle_t, int, csx_handle_t *) { return CSX_EINVAL; }
int csx_tri_info(csx_handle_t, int32_t *, int32_t *, int32_t *) { return CSX_EINVAL; }
int csx_tri_solve(csx_handle_t, csx_handle_t, int32_t) { return CSX_EINVAL; }
int csx_permute_vec(csx_handle_t, csx_handle_t, csx_handle_t, int32_t, int32_t, int) { return CSX_EINVAL; }
int csx_schol_host(int32_t, const int32_t *, const int32_t *, int32_t *, int32_t *) { return CSX_EINVAL; }
int csx_chol(csx_handle_t, const int32_t *, const int32_t *, const int32_t *, csx_handle_t *) { return CSX_EINVAL; }
int csx_cholsol_plan(csx_handle_t, const int32_t *, csx_handle_t *) { return CSX_EINVAL; }
int csx_cholsol_solve(csx_handle_t, csx_handle_t, int32_t) { return CSX_EINVAL; }
}
