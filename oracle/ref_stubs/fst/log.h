// TEST INFRASTRUCTURE ONLY: see fstlib.h in this directory.
#include "fst/fstlib.h"
