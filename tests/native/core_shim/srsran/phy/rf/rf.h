#pragma once
#include "core_standin.h"
