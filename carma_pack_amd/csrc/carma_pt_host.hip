// placeholder until the sampler lands
#include "carma_host.h"
namespace carma {
void pt_state_free(Ctx* c) { (void)c; }
}
