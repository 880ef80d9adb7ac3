// what the reference's arch sources include under this name: declarations only (include/visp/builders.h)
#pragma once
#include "../builders.h"
#include "../nn.h"
