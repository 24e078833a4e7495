// Library identity for the combat_hip C ABI (include/combat_hip.h).
#include "combat_hip.h"

#define COMBAT_ABI_VERSION 2

extern "C" const char *combat_version(void) { return "combat_hip gfx950 abi1"; }
extern "C" int combat_abi_version(void) { return COMBAT_ABI_VERSION; }
