extern "C" int ref_shim_version() { return 1; }
