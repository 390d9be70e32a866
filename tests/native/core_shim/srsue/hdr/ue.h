#pragma once
namespace srsue {}
namespace srsran {}
#include "srsran/standin_l2.h"   /* srsran::mac_pcap (a member of LTESniffer_Core, never used by it) */
