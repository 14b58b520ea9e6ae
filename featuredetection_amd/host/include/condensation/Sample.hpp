// Sample.hpp of the reference -- see condensation_all.hpp
#pragma once
#include "condensation/condensation_all.hpp"
