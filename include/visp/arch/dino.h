// Forwarder: the reference's arch sources include "visp/arch/dino.h"; the declarations live in visp/builders.h.
#pragma once
#include "../builders.h"
#include "../nn.h"
