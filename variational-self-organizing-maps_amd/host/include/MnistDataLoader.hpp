// Forwarding header (reference include name): everything lives in vsom_api.hpp.
#pragma once
#include "vsom_api.hpp"
