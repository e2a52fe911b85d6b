#define ILQG_FOR_DIMS(X) X(14, 3, 2)
