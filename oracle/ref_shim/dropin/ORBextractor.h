// oracle/_ref/liborbslam_dropin.so: the reference's own Frame.cc / ORBmatcher.cc compiled against THIS repository's drop-in
// ORBextractor class instead of the reference's include/ORBextractor.h.  The reference's headers include "ORBextractor.h" with quotes,
// which finds the file next to them first: this header is force-included (-include) ahead of everything and shares the reference's
// include guard, so theirs becomes empty.
#include "../../../include/ORBextractor.h"
