// libGenome/gnSequence.h -- forwarding header (genome::gnSequence slice used by the hot path).
#include "../libMems/mems_hip.h"
