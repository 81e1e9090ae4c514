#include "../cvlite.h"
