#pragma once
#include "imageprocessing/imageprocessing_all.hpp"
