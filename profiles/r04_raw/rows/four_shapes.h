#define AMT_MARCH_SHAPES(X) X(double, 1, 4, 1, 0, true, 16) X(double, 1, 3, 2, 0, true, 16) X(float, 2, 4, 1, 0, true, 16) X(float, 2, 3, 2, 0, true, 16)
