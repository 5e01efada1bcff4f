// TEST INFRASTRUCTURE ONLY (lane emulator): nothing of <hip/hip_fp16.h> is needed on the host side.
#pragma once
