// temporary: symbols that are declared in include/hiphase_gpu.h but not implemented yet
#include "hp_common.h"
extern "C" {
int hp_wfa_assign_batch(const hp_wfa_job*, size_t, uint64_t, uint64_t, hp_wfa_result*, uint8_t* const*, int) {
    hp::set_error("hp_wfa_assign_batch: not implemented yet");
    return HP_ERR_UNSUPPORTED;
}
int hp_edit_distance_batch(const hp_ed_pair*, size_t, uint64_t*, int) {
    hp::set_error("hp_edit_distance_batch: not implemented yet");
    return HP_ERR_UNSUPPORTED;
}
}
