// compat/webp/webp.h -- lets a program written against the reference's umbrella header
// (`#include "webp/webp.h"`, e.g. its command line tool webp.cpp:2) compile UNCHANGED against the
// MI355X mirror: put <repo>/include/compat and <repo>/include on the include path in place of the
// reference's tree.  The reference's header brings libpng and <stdlib.h>/<setjmp.h> in through
// its include chain (webp/webp.h:3-7, webp/platform.h:4-21) and its tool relies on that for
// read_png (webp.cpp:16-117), so this one does the same; everything else is the mirror.
#pragma once
#include <errno.h>
#include <math.h>
#include <setjmp.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <string>
#include <vector>

#include <png.h>

#include "webp_amd/webp.hpp"
