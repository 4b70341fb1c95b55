// TEST-ONLY forwarder to the cv stand-in of tests/cv_stub
#pragma once
#include <opencv2/opencv.hpp>
