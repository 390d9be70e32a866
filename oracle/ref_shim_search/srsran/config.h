/* see srsran/standin.h */
#pragma once
#include "srsran/standin.h"
