// all-steps drop-in build only: this repository's ORBmatcher class declaration in place of the reference's include/ORBmatcher.h
// (same public signatures + the nested Access type the friend line in MapPoint.h reaches).  -Iref_shim/dropin comes first, so every "ORBmatcher.h" of
// the build resolves here; the include guard is the reference's.
#include "../../../include/ORBmatcher.h"
