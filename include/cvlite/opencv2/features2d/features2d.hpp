#include "../../cvlite.h"
