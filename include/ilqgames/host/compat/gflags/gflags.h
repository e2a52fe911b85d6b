// Spelling of the flags include used by sources written for the reference.
#include <ilqgames/host/logging.hpp>
