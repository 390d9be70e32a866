/* see srsran/standin_l2.h */
#pragma once
#include "srsran/standin_l2.h"
