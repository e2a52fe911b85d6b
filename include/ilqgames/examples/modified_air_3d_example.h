#pragma once
#include <ilqgames/host/example_decl.hpp>
ILQGAMES_DECLARE_TOP_DOWN_EXAMPLE(ModifiedAir3DExample)
