// One environment kind's rollout kernels: compiled once per kind with -DPSH_KIND=<psh_env_kind value>.
#include "psh_rollout.h"

template int launch_rollout<PSH_KIND>(psh_ctx*, const psh_rollout_desc*, RolloutArgs&, hipStream_t);
