// libMems/ProgressiveAligner.h -- forwarding header: the hot-path surface lives in mems_hip.h (see its header note).
#include "mems_hip.h"
