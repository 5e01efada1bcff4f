// Identity of the CPU lane-emulator build of the kernels (tests only).
extern "C" const char* tzr_backend(void) { return "emu"; }
extern "C" int tzr_abi_version(void) { return 4; }
