// Drop-in for the reference header of the same name: the whole interface lives in mgcr_dropin.hpp.
#pragma once
#include "mgcr_dropin.hpp"
