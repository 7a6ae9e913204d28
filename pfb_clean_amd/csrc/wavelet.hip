// wavelet.hip -- psi / prox / primal-dual kernels (entry points stubbed until they land).
#include "common.hpp"
using namespace pfb;
extern "C" {
#define PFB_TODO(name) set_error(name ": not implemented yet"); return PFB_ERR_UNSUPPORTED
int pfb_psi_plan_create(int, int, int, int, const int*, const double*, int, int, pfb_psi_plan**) { PFB_TODO("pfb_psi_plan_create"); }
int pfb_psi_plan_destroy(pfb_psi_plan*) { return PFB_OK; }
int pfb_psi_plan_dims(const pfb_psi_plan*, int*, int*) { PFB_TODO("pfb_psi_plan_dims"); }
int pfb_psi_dot(pfb_psi_plan*, const void*, void*, void*) { PFB_TODO("pfb_psi_dot"); }
int pfb_psi_hdot(pfb_psi_plan*, const void*, void*, void*) { PFB_TODO("pfb_psi_hdot"); }
int pfb_dual_update(int, const void*, void*, const void*, double, double, int, size_t, void*, void*) { PFB_TODO("pfb_dual_update"); }
int pfb_prox_21m(int, const void*, void*, const void*, double, double, int, size_t, void*) { PFB_TODO("pfb_prox_21m"); }
int pfb_pd_primal_update(int, const void*, const void*, const void*, double, int, int, size_t, void*, double*, double*, void*) { PFB_TODO("pfb_pd_primal_update"); }
}
