// eval path: kNN-blended field evaluation (models.py:347-405) -- placeholder until the fused kernel lands.
#include "ngm_device.h"
#include "ngm_launch.h"
int ngm_launch_knn(const ngm_field_cfg*, const ngm_params*, int, int64_t, const float*, const float*, const float*, int,
                   float, float, float*, hipStream_t) {
  return NGM_E_UNSUPPORTED;
}
