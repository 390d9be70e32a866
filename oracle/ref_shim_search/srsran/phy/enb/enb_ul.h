/* see srsran/standin_ul.h */
#pragma once
#include "srsran/standin_ul.h"
