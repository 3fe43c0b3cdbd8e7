/* file_buffer.h -- byte stream: see dclib_boundary.h (this name exists so that sources written for the reference find it) */
#include "dclib_boundary.h"
