// Same include path as the reference; the mirrored declarations live in one header.
#include <ilqgames/host/api.hpp>
