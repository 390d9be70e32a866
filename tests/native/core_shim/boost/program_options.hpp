#pragma once
namespace boost { namespace program_options {} }
