#pragma once
#include "srsran/asn1/rrc/standin_rrc.h"
