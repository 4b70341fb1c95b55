// TEST-ONLY forwarder (see slam_stub_types.h)
#pragma once
#include "../slam_stub_types.h"
