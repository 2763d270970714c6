/* stb_ref.c -- builds the reference's own vendored stb headers (utils/stb_image.h, utils/stb_image_write.h),
 * included from where they lie under /root/reference (-I$(REFERENCE)), into oracle/_ref/libstb_ref.so.
 * Test infrastructure only: the real reference for lights.cpp:34 (stbi_loadf) and canvas.cpp:102
 * (stbi_write_tga).  Contains no reference source. */
#define STB_IMAGE_IMPLEMENTATION
#include "utils/stb_image.h"
#define STB_IMAGE_WRITE_IMPLEMENTATION
#include "utils/stb_image_write.h"
