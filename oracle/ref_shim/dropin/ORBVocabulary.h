// all-steps drop-in build only: this repository's ORBVocabulary class in place of the reference's typedef of the DBoW2
// template; force-included ahead of everything, shares the reference's include guard.  BowVector / FeatureVector stay the reference's classes.
#include "../../../include/ORBVocabulary.h"
