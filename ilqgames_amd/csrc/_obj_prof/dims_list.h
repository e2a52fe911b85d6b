#define ILQG_FOR_DIMS(X) X(24, 4, 2)
