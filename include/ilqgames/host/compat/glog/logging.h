// Spelling of the logging include used by sources written for the reference.
#include <ilqgames/host/logging.hpp>
