// mds_platform_rt.h - runtime header of the product build (HIP).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <atomic>
