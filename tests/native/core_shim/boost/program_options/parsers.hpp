#pragma once
#include "boost/program_options.hpp"
