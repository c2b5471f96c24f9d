// Drop-in for the reference's stan/gptools/util.stan: `#include gptools/util.stan` alone suffices.
//
//  * Real-FFT layer (reference util.stan:255-330): DECLARATIONS only; the bodies live in libgptools_hip.so and are
//    bound by include/gptools_hip.hpp (stanc --allow-undefined).
//  * Everything else the reference's file offers (debug assertions util.stan:9-250, gp_conditional_loc_scale
//    util.stan:344-350) is not on the accelerated path and stays ordinary Stan, restated here with the reference's
//    signatures, tolerances (rtol 1e-6, atol 0, floor 1e-15) and reject messages so that models calling them compile
//    and behave unchanged.

// ---- accelerated: declarations ---------------------------------------------------------------------
complex_vector rfft(vector y);
complex_vector expand_rfft(complex_vector y, int n);
vector inv_rfft(complex_vector z, int n);
complex_matrix rfft2(matrix y);
matrix inv_rfft2(complex_matrix z, int m);

// ---- plain Stan: scalar checks -----------------------------------------------------------------------
void assert_equal(int actual, int desired) {
    if (actual != desired) reject(actual, " is not equal to ", desired);
}

// |actual - desired| <= max(atol + rtol |desired|, 1e-15)
int is_close(real actual, real desired, real rtol, real atol) {
    real allowed = fmax(atol + rtol * abs(desired), 1e-15);
    return abs(actual - desired) <= allowed;
}

void assert_close(real actual, real desired, real rtol, real atol) {
    if (is_close(actual, desired, rtol, atol) == 0) reject(actual, " is not close to ", desired);
}
void assert_close(real actual, real desired) {
    assert_close(actual, desired, 1e-6, 0);
}

// 1 when neither part is NaN or infinite (a real argument promotes to complex)
int is_finite(complex x) {
    array[2] real parts = {get_real(x), get_imag(x)};
    for (k in 1:2) {
        if (is_nan(parts[k]) || is_inf(parts[k])) return 0;
    }
    return 1;
}

// ---- vectors ----------------------------------------------------------------------------------------
void assert_close(vector actual, vector desired, real rtol, real atol) {
    if (size(actual) != size(desired)) {
        reject("number of elements are not equal: size(desired)=", size(desired), "; size(actual)=", size(actual));
    }
    for (i in 1:size(desired)) {
        if (is_close(actual[i], desired[i], rtol, atol) == 0) {
            reject(actual[i], " is not close to ", desired[i], " at position ", i);
        }
    }
}
void assert_close(vector actual, vector desired) {
    assert_close(actual, desired, 1e-6, 0);
}
void assert_close(vector actual, real desired, real rtol, real atol) {
    assert_close(actual, rep_vector(desired, size(actual)), rtol, atol);
}
void assert_close(vector actual, real desired) {
    assert_close(actual, desired, 1e-6, 0);
}

int is_finite(vector x) {
    int ok = 1;
    for (i in 1:size(x)) {
        if (is_finite(x[i]) == 0) ok = 0;
    }
    return ok;
}
void assert_finite(vector x) {
    for (i in 1:size(x)) {
        if (is_finite(x[i]) == 0) reject(x[i], " at index ", i, " is not finite");
    }
}

// ---- matrices ---------------------------------------------------------------------------------------
int is_finite(matrix x) {
    return is_finite(to_vector(x));
}

void print_matrix(complex_matrix x) {
    print("matrix with ", rows(x), " rows and ", cols(x), " columns");
    for (r in 1:rows(x)) print(x[r]);
}

void assert_close(matrix actual, matrix desired, real rtol, real atol) {
    array[2] int want = dims(desired);
    array[2] int got = dims(actual);
    if (got[1] != want[1]) {
        reject("number of rows are not equal: dims(desired)[1]=", want[1], "; size(actual)[1]=", got[1]);
    }
    if (got[2] != want[2]) {
        reject("number of columns are not equal: dims(desired)[2]=", want[2], "; size(actual)[2]=", got[2]);
    }
    for (i in 1:want[1]) {
        for (j in 1:want[2]) {
            if (is_close(actual[i, j], desired[i, j], rtol, atol) == 0) {
                reject(actual[i, j], " is not close to ", desired[i, j], " at row ", i, ", column ", j);
            }
        }
    }
}
void assert_close(matrix actual, matrix desired) {
    assert_close(actual, desired, 1e-6, 0);
}
void assert_close(matrix actual, real desired, real rtol, real atol) {
    assert_close(actual, rep_matrix(desired, rows(actual), cols(actual)), rtol, atol);
}
void assert_close(matrix actual, real desired) {
    assert_close(actual, desired, 1e-6, 0);
}

// ---- complex vectors: real and imaginary parts are held to the same tolerances -------------------------
void assert_close(complex_vector actual, complex_vector desired, real rtol, real atol) {
    if (size(actual) != size(desired)) {
        reject("number of elements are not equal: size(desired)=", size(desired), "; size(actual)=", size(actual));
    }
    assert_close(get_real(actual), get_real(desired), rtol, atol);
    assert_close(get_imag(actual), get_imag(desired), rtol, atol);
}
void assert_close(complex_vector actual, complex_vector desired) {
    assert_close(actual, desired, 1e-6, 0);
}
void assert_close(complex_vector actual, complex desired, real rtol, real atol) {
    assert_close(actual, rep_vector(desired, size(actual)), rtol, atol);
}
void assert_close(complex_vector actual, complex desired) {
    assert_close(actual, desired, 1e-6, 0);
}

// ---- conditional normal (used by the reference's graph path; kept so that the name keeps resolving) -------
// Location and scale of x1 | x2 = y for jointly normal (x1, x2) with zero mean, var(x1) = cov11, cov(x2, x1) = cov21,
// cov(x2) = cov22; with nothing to condition on: (0, sqrt(cov11)).
vector gp_conditional_loc_scale(vector y, real cov11, vector cov21, matrix cov22) {
    vector[2] result;
    if (size(y) == 0) {
        result[1] = 0;
        result[2] = sqrt(cov11);
    } else {
        vector[size(y)] weights = mdivide_left_spd(cov22, cov21);
        result[1] = dot_product(weights, y);
        result[2] = sqrt(cov11 - dot_product(weights, cov21));
    }
    return result;
}
